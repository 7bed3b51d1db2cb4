#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03af; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lite" > $out/tests.log 2>&1; tail -2 $out/tests.log
python bench.py --lite --steps 10 --warmup 3 > $out/bench_sio2_10k_setA_lite.json 2>/dev/null; python -c "
import json; d = json.loads(open('$out/bench_sio2_10k_setA_lite.json').read().strip().splitlines()[-1]); print('lite', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('accuracy', {}).get('rel_max'))"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03z; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "vs_twin or separate_kernels" > $out/tests.log 2>&1; tail -12 $out/tests.log | cut -c1-250
python bench.py --lite --steps 10 --warmup 3 > $out/bench_sio2_10k_setA_lite.json 2> $out/bench.err; python -c "
import json; d = json.loads(open('$out/bench_sio2_10k_setA_lite.json').read().strip().splitlines()[-1]); print('lite', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('accuracy'), d.get('cpu_baseline'))"

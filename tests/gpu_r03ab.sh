#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03ab; mkdir -p $out
for i in 1 2; do python tests/bench_tp.py --nodes 16384 2>&1 | grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('tp_is setA', d['kernel'], round(d['ms'],3), d['checksum'])"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lite or message_pack or sio2" > $out/tests.log 2>&1; tail -3 $out/tests.log
python bench.py --lite --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sio2_10k_setA_lite.json 2> $out/bench.err; python -c "
import json; d = json.loads(open('$out/bench_sio2_10k_setA_lite.json').read().strip().splitlines()[-1]); print('lite', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d.get('accuracy'))"

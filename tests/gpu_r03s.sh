#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03s; mkdir -p $out
python tests/bench_wgrad.py --splits 8,16,32,64 > $out/wgrad_splits.log 2>&1; grep '^{' $out/wgrad_splits.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['E'], d['nsplit'], round(d['ms'], 3), round(d['issued_TF'],1), d['checksum'], d['gs'])"
python tests/bench_wgrad.py --edges 131072 --splits 32 2>&1 | grep '^{' | cut -c1-200
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or training or weight or gradient or refresh or loss" > $out/tests.log 2>&1; tail -4 $out/tests.log
python tests/bench_training.py --workload si512 --steps 4 > $out/train_si512_fused.log 2>&1; tail -2 $out/train_si512_fused.log

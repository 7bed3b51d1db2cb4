#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03ai; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or training or weight or gradient or refresh or loss or repack" > $out/tests.log 2>&1; tail -3 $out/tests.log
python tests/bench_training.py --workload si512 --steps 5 > $out/train_si512.log 2>&1; tail -2 $out/train_si512.log
python tests/bench_training.py --workload si64 --steps 5 > $out/train_si64.log 2>&1; tail -1 $out/train_si64.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03v; mkdir -p $out
python tests/bench_wgrad.py --splits 32,64 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['E'], d['nsplit'], round(d['ms'], 3), round(d['issued_TF'],1), d['checksum'], d['gs'])"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or training or weight or gradient or refresh or loss" > $out/tests.log 2>&1; tail -2 $out/tests.log
python tests/bench_training.py --workload si512 --steps 4 > $out/train_si512_fused.log 2>&1; tail -2 $out/train_si512_fused.log
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python tests/bench_training.py --workload si512 --steps 3 > $out/train_si512.log 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/training_si512_kernel_stats.csv
rm -rf $out/prof

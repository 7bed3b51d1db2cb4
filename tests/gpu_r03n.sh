#!/bin/bash
# round-3: fused weight-gradient kernel -- parity (all backward / training tests go through it by default) + training step time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03n; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward or training or weight or gradient or refresh or loss" > $out/tests.log 2>&1; tail -6 $out/tests.log
python tests/bench_training.py --workload si512 --steps 4 > $out/train_si512_fused.log 2>&1; tail -3 $out/train_si512_fused.log
HG_WGRAD=rows python tests/bench_training.py --workload si512 --steps 4 > $out/train_si512_rows.log 2>&1; tail -3 $out/train_si512_rows.log
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python tests/bench_training.py --workload si512 --steps 3 > $out/train_si512.log 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/training_si512_kernel_stats.csv
rm -rf $out/prof
head -24 $out/training_si512_kernel_stats.csv | cut -c1-150

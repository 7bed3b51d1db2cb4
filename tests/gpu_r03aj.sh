#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03aj; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "head or sio2 or fixture" > $out/tests.log 2>&1; tail -2 $out/tests.log
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-accuracy > $out/bench_profiled.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/sio2_10k_kernel_stats.csv; rm -rf $out/prof
grep -E "ham_readout|row_program" $out/sio2_10k_kernel_stats.csv | cut -c1-40,180-260

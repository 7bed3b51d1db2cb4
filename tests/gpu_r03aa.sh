#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03aa; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python bench.py --lite --steps 3 --warmup 1 --no-cpu-baseline --no-accuracy > $out/bench_profiled.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/lite_kernel_stats.csv; rm -rf $out/prof
head -10 $out/lite_kernel_stats.csv | cut -c1-160

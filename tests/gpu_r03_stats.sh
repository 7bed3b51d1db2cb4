#!/bin/bash
# end of round 3: rocprofv3 kernel stats of the benchmarked command on the final build (-> profiles/r03_sio2_10k_kernel_stats.*)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03stats; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-accuracy --no-mfma-probe > $out/bench_profiled.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/sio2_10k_kernel_stats.csv
rm -rf $out/prof
head -12 $out/sio2_10k_kernel_stats.csv | cut -c1-160

cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02t; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python tests/bench_training.py --workload si512 --steps 3 > $out/train_si512.log 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/training_si512_kernel_stats.csv
rm -rf $out/prof
tail -2 $out/train_si512.log
head -25 $out/training_si512_kernel_stats.csv | cut -c1-160

cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02c; mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench_sio2_10k_setA.json 2> $out/bench.err
python bench.py --steps 200 --warmup 10 --workload si2 --no-cpu-baseline > $out/bench_si2_setA.json 2>> $out/bench.err
python bench.py --steps 30 --warmup 5 --workload si512 --irreps B > $out/bench_si512_setB.json 2>> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_profiled.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/sio2_10k_kernel_stats.csv
rm -rf $out/prof
ls $out

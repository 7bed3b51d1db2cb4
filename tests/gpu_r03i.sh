cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03i; mkdir -p $out
timeout 600 python bench.py --workload uni8 --steps 5 --warmup 2 > $out/bench_uni8.json 2> $out/bench_uni8.err; tail -5 $out/bench_uni8.err
cat $out/bench_uni8.json

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03a; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "static_stream or sio2_setA" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
for st in 1 0; do HG_ST=$st timeout 200 python tests/bench_tp.py --nodes 16384 --reps 8 --tag A_st$st 2>&1 | tail -1; done > $out/bench_tp.jsonl
for st in 1 0; do HG_ST=$st timeout 200 python tests/bench_tp.py --irreps B --nodes 16384 --reps 8 --tag B_st$st 2>&1 | tail -1; done >> $out/bench_tp.jsonl
HG_ST=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_st1.json 2> $out/bench_st1.err
HG_ST=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/bench_st0.json 2> $out/bench_st0.err
tail -3 $out/pytest.log; cat $out/bench_tp.jsonl; cat $out/bench_st1.json $out/bench_st0.json

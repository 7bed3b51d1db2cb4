cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03h; mkdir -p $out
timeout 600 python bench.py --steps 10 --warmup 3 > $out/bench_sio2_10k_setA.json 2> $out/bench.err; tail -3 $out/bench.err
timeout 600 python bench.py --workload uni8 --steps 5 --warmup 2 > $out/bench_uni8.json 2> $out/bench_uni8.err; tail -5 $out/bench_uni8.err
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bench_script_two_rank or batch_of_crystals or uni_hamgnn" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
cat $out/bench_sio2_10k_setA.json; cat $out/bench_uni8.json

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "static_stream or sio2_setA" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
for st in 1 0; do HG_ST=$st timeout 200 python tests/bench_tp.py --nodes 16384 --reps 8 --tag A_st$st 2>&1 | tail -1; done > $out/bench_tp.jsonl
for st in 1 0; do HG_ST=$st timeout 200 python tests/bench_tp.py --irreps B --nodes 16384 --reps 8 --tag B_st$st 2>&1 | tail -1; done >> $out/bench_tp.jsonl
HG_PROF=1 HG_ST=1 HG_LIB_PATH=hamgnn_amd/lib/variants/lib_prof.so timeout 200 python tests/bench_tp.py --nodes 16384 --reps 5 --tag prof_st1 2>&1 | tail -2 >> $out/bench_tp.jsonl
tail -3 $out/pytest.log; cat $out/bench_tp.jsonl

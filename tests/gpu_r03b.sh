cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03b; mkdir -p $out
for st in 1 0; do HG_PROF=1 HG_ST=$st HG_LIB_PATH=hamgnn_amd/lib/variants/lib_prof.so timeout 200 python tests/bench_tp.py --nodes 16384 --reps 5 --tag prof_st$st 2>&1 | tail -2; done > $out/prof.jsonl
cat $out/prof.jsonl

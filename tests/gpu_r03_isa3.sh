#!/bin/bash
# tp_wgrad with sched_group_barrier (reads batched ahead of the MFMAs) vs the compiler's order; parity of the backward tests on the new build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03isa3; mkdir -p $out
V=hamgnn_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wgrad" > $out/tests.log 2>&1; tail -2 $out/tests.log
for i in 1 2; do
  timeout 200 python tests/bench_wgrad.py --splits 32,64 --reps 5 --tag sgb >> $out/wg.jsonl 2>> $out/err.log
  HG_LIB_PATH=$V/lib_wgnosgb.so timeout 200 python tests/bench_wgrad.py --splits 32,64 --reps 5 --tag nosgb >> $out/wg.jsonl 2>> $out/err.log
  HG_LIB_PATH=$V/lib_wgp2c.so timeout 200 python tests/bench_wgrad.py --splits 32,64 --reps 5 --tag p2c >> $out/wg.jsonl 2>> $out/err.log
done
cat $out/wg.jsonl | cut -c1-300

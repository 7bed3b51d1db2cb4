#!/bin/bash
# r4 batch B: new default tp_is (packed cf by DPP + lazily re-read resident hidden rows) vs r3 (variant "r3" = git d231463 build), phase profile, parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04b; mkdir -p $out; rm -f $out/bench.log
V=hamgnn_amd/lib/variants
for rep in 1 2; do
  for n in $(cat $V/list.txt); do
    HG_LIB_PATH=$PWD/$V/lib_$n.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag $n 2>&1 | tail -1 >> $out/bench.log
  done
done
HG_LIB_PATH=$PWD/$V/lib_new.so timeout 120 python tests/bench_tp.py --irreps B --nodes 16384 --reps 8 --tag newB 2>&1 | tail -1 >> $out/bench.log
HG_BENCH_LDS=163840 HG_LIB_PATH=$PWD/$V/lib_new.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag new_1wg 2>&1 | tail -1 >> $out/bench.log
HG_PROF=1 HG_LIB_PATH=$PWD/$V/lib_prof.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 3 --tag prof 2>&1 | tail -2 >> $out/bench.log
cat $out/bench.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "message_pack or sio2_setA or si2_default or backbone_golden" > $out/tests.log 2>&1; tail -3 $out/tests.log

#!/bin/bash
# r4 batch A: item-level variants of tp_is (packed cf by DPP, resident hidden rows, early A2, dual accumulator chains) + 1-WG/CU diagnostic
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04a; mkdir -p $out
V=hamgnn_amd/lib/variants
for rep in 1 2; do
for n in base cfp dual cfpd hb cfphb cfpea2 cfphbd; do
  HG_LIB_PATH=$PWD/$V/lib_$n.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag $n 2>&1 | tail -1 >> $out/bench.log
done
done
HG_BENCH_LDS=163840 HG_LIB_PATH=$PWD/$V/lib_base.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag base_1wg 2>&1 | tail -1 >> $out/bench.log
HG_BENCH_LDS=163840 HG_LIB_PATH=$PWD/$V/lib_cfphbd.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag cfphbd_1wg 2>&1 | tail -1 >> $out/bench.log
cat $out/bench.log
HG_LIB_PATH=$PWD/$V/lib_cfphbd.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "message_pack_block_golden or message_pack_random or sio2_setA or data_gradient_default" > $out/tests_cfphbd.log 2>&1; tail -3 $out/tests_cfphbd.log
HG_LIB_PATH=$PWD/$V/lib_cfpea2.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "message_pack_block_golden or sio2_setA" > $out/tests_cfpea2.log 2>&1; tail -3 $out/tests_cfpea2.log

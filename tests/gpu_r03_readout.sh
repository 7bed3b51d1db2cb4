#!/bin/bash
# software-pipelined read-out kernel vs the r3 build (variants/lib_headbase.so): parity of the head tests, ms per 822 k rows
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03readout; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "head or readout or ham or full_model or soc or golden" > $out/tests.log 2>&1; tail -2 $out/tests.log
for i in 1 2; do
  timeout 200 python tests/bench_readout.py --tag new >> $out/ro.jsonl 2>> $out/err.log
  HG_LIB_PATH=hamgnn_amd/lib/variants/lib_headbase.so timeout 200 python tests/bench_readout.py --tag base >> $out/ro.jsonl 2>> $out/err.log
done
cat $out/ro.jsonl; tail -3 $out/err.log

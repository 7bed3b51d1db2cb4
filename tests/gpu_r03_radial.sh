#!/bin/bash
# radial hidden layers, two tiles per step vs the r3 build (variants/lib_auxbase.so)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03radial; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "radial or backbone or conv or golden or fixture" > $out/tests.log 2>&1; tail -2 $out/tests.log
for i in 1 2; do
  timeout 200 python tests/bench_radial.py --tag new >> $out/r.jsonl 2>> $out/err.log
  HG_LIB_PATH=hamgnn_amd/lib/variants/lib_auxbase.so timeout 200 python tests/bench_radial.py --tag base >> $out/r.jsonl 2>> $out/err.log
done
timeout 200 python tests/bench_radial.py --rows 44033 --tag new_small >> $out/r.jsonl 2>> $out/err.log
cat $out/r.jsonl; tail -3 $out/err.log | grep -v amdgpu.ids

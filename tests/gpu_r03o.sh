#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r03o; mkdir -p $out
python tests/bench_wgrad.py --splits 6,16,32,64,128 > $out/wgrad_splits.log 2>&1; cat $out/wgrad_splits.log | tail -6
python tests/bench_wgrad.py --edges 131072 --splits 16,32,64 >> $out/wgrad_splits.log 2>&1; tail -3 $out/wgrad_splits.log

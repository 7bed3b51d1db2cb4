#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04l; mkdir -p $out
timeout 900 python tests/bench_training.py --workload si512 --steps 5 --profile > $out/si512.log 2>&1; grep "^step" $out/si512.log; grep -A 60 "cumulative" $out/si512.log | cut -c1-180 | head -75

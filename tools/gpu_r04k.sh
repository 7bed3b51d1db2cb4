#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04k; mkdir -p $out
timeout 600 python tests/bench_training_graph.py --workload si64 --steps 5 > $out/si64.log 2>&1; tail -5 $out/si64.log
timeout 900 python tests/bench_training_graph.py --workload si512 --steps 5 > $out/si512.log 2>&1; tail -5 $out/si512.log

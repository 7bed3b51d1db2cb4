#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r03r; mkdir -p $out
python tests/bench_training.py --workload si512 --steps 4 --profile > $out/train_profile.log 2>&1; grep -n "step " $out/train_profile.log | tail -2

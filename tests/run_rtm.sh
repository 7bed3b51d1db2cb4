#!/bin/bash
# planner row-tile table sweep (HG_RTM = max row tiles per item, indexed by MM) with the one-launch micro-benchmark
cd "$(dirname "$0")/.."
for t in ${HG_RTM_LIST:-"4,4,4,3,2,2,1"}; do
  HG_RTM=$t timeout 120 python tests/bench_tp.py --reps 5 --tag "$t" "$@" 2>&1 | tail -1
done | tee gpurun_out/rtm.jsonl

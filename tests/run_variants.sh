#!/bin/bash
# times every kernel variant in hamgnn_amd/lib/variants with the one-launch micro-benchmark (run on the GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for so in hamgnn_amd/lib/variants/lib_*.so; do
  HG_LIB_PATH=$so timeout 120 python tests/bench_tp.py --reps 5 "$@" 2>&1 | tail -1
done | tee gpurun_out/variants.jsonl

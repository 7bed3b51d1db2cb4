#!/bin/bash
# builds kernel variants of libhamgnn_hip.so into gpurun_out/variants/ for A/B timing on the GPU box:  name:"extra hipcc flags"
set -e
cd "$(dirname "$0")/../hamgnn_amd/csrc"
mkdir -p ../lib/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I../../include $flags -shared tp_fused.hip tp_is.hip aux_kernels.hip head.hip attention.hip linear.hip -o ../lib/variants/lib_$name.so &
done
wait
ls -la ../lib/variants/

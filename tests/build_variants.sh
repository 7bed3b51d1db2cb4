#!/bin/bash
# builds kernel variants of libhamgnn_hip.so into hamgnn_amd/lib/variants/ for A/B timing on the GPU box:  name:"extra hipcc flags"
# (only the edge kernels (tp_is, tp_st, tp_wgrad) are recompiled with the flags; the other objects come from the regular build)
set -e
cd "$(dirname "$0")/../hamgnn_amd/csrc"
make -j8 > /dev/null
mkdir -p ../lib/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( for f in tp_is tp_st tp_wgrad; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I../../include $flags -c $f.hip -o ../lib/variants/${f}_$name.o; done
    hipcc --offload-arch=gfx950 -shared -fPIC ../lib/tp_fused.o ../lib/variants/tp_is_$name.o ../lib/variants/tp_st_$name.o ../lib/variants/tp_wgrad_$name.o ../lib/aux_kernels.o ../lib/head.o ../lib/attention.o ../lib/linear.o ../lib/rowprog.o -o ../lib/variants/lib_$name.so
    rm -f ../lib/variants/tp_is_$name.o ../lib/variants/tp_st_$name.o ../lib/variants/tp_wgrad_$name.o ) &
done
wait
ls -la ../lib/variants/

#!/bin/bash
# builds kernel variants of libhamgnn_hip.so into hamgnn_amd/lib/variants/ for A/B timing on the GPU box:  name:"extra hipcc flags"
# Only the files named in HG_VARIANT_FILES (default: tp_is) are recompiled with the flags; the other objects come from the regular build.
#   tools/build_variants.sh base: cfp:"-DHG_CFP"        ->  hamgnn_amd/lib/variants/lib_base.so, lib_cfp.so  (select with HG_LIB_PATH)
set -e
cd "$(dirname "$0")/../hamgnn_amd/csrc"
make -j8 > /dev/null
mkdir -p ../lib/variants
FILES=${HG_VARIANT_FILES:-tp_is}
ALL="tp_fused tp_is tp_wgrad aux_kernels head attention linear linear_wgrad rowprog block_gemm corr3"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( objs=""
    for f in $ALL; do
      if [[ " $FILES " == *" $f "* ]]; then
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value ${HG_VARIANT_NOPK--Xclang -target-feature -Xclang -packed-fp32-ops} -I../../include $flags -c $f.hip -o ../lib/variants/${f}_$name.o
        objs="$objs ../lib/variants/${f}_$name.o"
      else
        objs="$objs ../lib/$f.o"
      fi
    done
    hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../lib/variants/lib_$name.so
    rm -f ../lib/variants/*_$name.o ) &
  while [ "$(jobs -r | wc -l)" -ge 6 ]; do sleep 0.5; done
done
wait
ls -la ../lib/variants/

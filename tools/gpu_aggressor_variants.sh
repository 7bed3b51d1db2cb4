# r6 (profiles/r06_tp_is.md section 8): kernel-variant libraries as the victim of the MFMA-only aggressor:  bash tools/gpu_aggressor_variants.sh <tag> "<variants>" [modes]
cd /tmp && export TMPDIR=/tmp; out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06av}; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -shared -fPIC $GRAFT_REPO_ROOT/tests/csrc/xdl_aggressor.hip -o /tmp/libxdl_aggressor.so || exit 1
cd $GRAFT_REPO_ROOT
for v in $2; do
  if [ "$v" = "default" ]; then lib=$GRAFT_REPO_ROOT/hamgnn_amd/lib/libhamgnn_hip.so; else lib=$GRAFT_REPO_ROOT/hamgnn_amd/lib/variants/lib_$v.so; fi
  echo "== victim library: $v" | tee -a $out/variants.log
  HG_LIB_PATH=$lib timeout 300 python tools/gpu_aggressor.py /tmp/libxdl_aggressor.so --grid 256 --launches 5 --modes ${3:-0,1} 2>&1 | grep "^{" | cut -c1-330 | tee -a $out/variants.log
done

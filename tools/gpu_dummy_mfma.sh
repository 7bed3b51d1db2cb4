# r6 (profiles/r06_tp_is.md section 8): the SHIPPED kernel + MFMAs whose result nobody reads (variants of tools/build_variants.sh: xdl / smfma = -DK_XDL_DUMMY / -DK_SMFMA_DUMMY,
# *_ls = + -DIS_DEAL_ALL) against the default library's result, bit for bit.   bash tools/gpu_dummy_mfma.sh <tag> "<variants>" [forwards] [workload]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06dm}; mkdir -p $out
wl=${4:-sio2_10k}
timeout 400 python tools/gpu_deal_all.py save /tmp/h_$wl.pt --workload $wl 2>/dev/null | tail -1 | tee -a $out/dummy.log
for v in $2; do
  HG_LIB_PATH=$GRAFT_REPO_ROOT/hamgnn_amd/lib/variants/lib_$v.so timeout 600 python tools/gpu_deal_all.py check /tmp/h_$wl.pt --workload $wl --forwards ${3:-20} 2>&1 | tail -1 | tee -a $out/dummy.log
done

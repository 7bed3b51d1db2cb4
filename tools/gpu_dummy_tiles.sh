# r6 (profiles/r06_tp_is.md section 8): wrong tiles of one launch per variant library:  bash tools/gpu_dummy_tiles.sh <tag> "<variants>" [launches]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06dt}; mkdir -p $out
timeout 300 python tools/gpu_dummy_tiles.py save /tmp/o.pt 2>/dev/null | tail -1 | tee -a $out/tiles.log
for v in $2; do
  HG_LIB_PATH=$GRAFT_REPO_ROOT/hamgnn_amd/lib/variants/lib_$v.so timeout 300 python tools/gpu_dummy_tiles.py check /tmp/o.pt --launches ${3:-6} 2>&1 | tail -1 | tee -a $out/tiles.log
done

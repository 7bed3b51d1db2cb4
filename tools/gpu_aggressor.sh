# r6 (profiles/r06_tp_is.md section 8): the unmodified shipped library next to a separate MFMA-only kernel on a side stream
cd /tmp && export TMPDIR=/tmp; out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06ag}; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -shared -fPIC $GRAFT_REPO_ROOT/tests/csrc/xdl_aggressor.hip -o /tmp/libxdl_aggressor.so || exit 1
cd $GRAFT_REPO_ROOT
if [ "$4" = "2" ]; then timeout 800 python tools/gpu_aggressor2.py /tmp/libxdl_aggressor.so 2>&1 | grep -v Warning | tail -40 | tee -a $out/aggressor2.log; exit 0; fi
for grid in ${2:-1024}; do timeout 600 python tools/gpu_aggressor.py /tmp/libxdl_aggressor.so --grid $grid --launches ${3:-6} 2>&1 | grep -v Warning | tail -7 | tee -a $out/aggressor.log; done

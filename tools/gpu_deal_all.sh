# r6: the shipped kernel with its waves in lock step (variant -DIS_DEAL_ALL: tools/build_variants.sh dealall:"-DIS_DEAL_ALL") against the default library, bit for bit
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06da}; mkdir -p $out
for wl in sio2_10k mos2_1200; do
  timeout 400 python tools/gpu_deal_all.py save /tmp/h_$wl.pt --workload $wl 2>/dev/null | tail -1 | tee -a $out/deal_all.log
  HG_LIB_PATH=$GRAFT_REPO_ROOT/hamgnn_amd/lib/variants/lib_dealall.so timeout 600 python tools/gpu_deal_all.py check /tmp/h_$wl.pt --workload $wl --forwards ${2:-40} 2>/dev/null | tail -1 | tee -a $out/deal_all.log
  timeout 600 python tools/gpu_deal_all.py check /tmp/h_$wl.pt --workload $wl --forwards 10 2>/dev/null | tail -1 | tee -a $out/deal_all.log
done
for lib in hamgnn_amd/lib/libhamgnn_hip.so hamgnn_amd/lib/variants/lib_dealall.so; do
  HG_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy --no-complete-pass 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib', round(r['value']), 'edges/s', round(r['ms_per_step'],2), 'ms')" | tee -a $out/deal_all.log
done

#!/bin/bash
# late r5: split launches of small crystals with the heaviest output segments on two workgroups each (plan.split_heavy_segments, SEG_ATOMIC epilogue): same-call A/B + parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r05ss}; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixture or golden or default_irreps or si2 or uni or oracle or unread or structural or attribute or front_door or corr or reproducible or soc" > $out/tests.log 2>&1
echo "tests exit $?"; tail -3 $out/tests.log
for rep in 1 2; do
for env in "HG_SPLIT_SEGMENTS=0" "HG_SPLIT_SEGMENTS=1"; do
  for wl in si2 sio2_24; do
    env $env timeout 200 python bench.py --steps 100 --warmup 10 --workload $wl --no-cpu-baseline --no-accuracy --no-mfma-probe 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env', '$wl', round(r['value']), 'edges/s', round(r['ms_per_step'],3), 'ms  median', round(r.get('ms_per_step_median',0),3), r['roofline']['launch_ms_by_position_in_step'])"
  done
done
done
for env in "HG_SPLIT_SEGMENTS=0" "HG_SPLIT_SEGMENTS=1"; do
  env $env timeout 300 python bench.py --workload uni8 --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=r.get('modes') or r['config'].get('modes'); print('$env', 'uni8 batched', round(r['value']), 'per crystal', round(m['per_crystal_eager']['edges_per_s']), [round(x,2) for x in m['per_crystal_graph_replay']['latency_ms_per_crystal']])"
done

#!/bin/bash
# late r5: the smallest crystals with the PHASES of a tile on separate workgroups (plan.is_schedule "phases", atomic-add epilogue): same-call A/B + parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r05pp}; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixture or golden or default_irreps or si2 or uni or oracle or unread or structural or attribute or front_door or corr or soc or backward or training" > $out/tests.log 2>&1
echo "tests exit $?"; tail -3 $out/tests.log
for rep in 1 2; do
for env in "HG_PHASE_PARTS=0" "HG_PHASE_PARTS=1" "HG_PHASE_PARTS_TILES=4096"; do
  for wl in si2 sio2_24; do
    env $env timeout 200 python bench.py --steps 100 --warmup 10 --workload $wl --no-cpu-baseline --no-accuracy --no-mfma-probe 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env', '$wl', round(r['value']), 'edges/s', round(r['ms_per_step'],3), 'ms  median', round(r.get('ms_per_step_median',0),3), r['roofline']['launch_ms_by_position_in_step'])"
  done
done
done

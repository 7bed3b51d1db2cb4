#!/bin/bash
# late r5 experiments on the smallest crystals (off by default): HG_PHASE_PARTS=1 (the phases of a tile on separate workgroups, every workgroup holds all tiles) and
# HG_PHASE_PARTS=2d3 (one workgroup per (output segment, third of its phases)); both add their tiles into zero-filled rows.  Same-call A/B + parity.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r05pp}; mkdir -p $out
HG_PHASE_PARTS=${2:-2d3} timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixture or golden or default_irreps or si2 or oracle or unread or structural" > $out/tests.log 2>&1
echo "tests ($2) exit $?"; tail -2 $out/tests.log
for rep in 1 2; do
for env in "HG_PHASE_PARTS=0" "HG_PHASE_PARTS=2d2" "HG_PHASE_PARTS=2d3" "HG_PHASE_PARTS=2d4"; do
  for wl in si2; do
    env $env timeout 200 python bench.py --steps 100 --warmup 10 --workload $wl --no-cpu-baseline --no-accuracy --no-mfma-probe 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env', '$wl', round(r['value']), 'edges/s', round(r['ms_per_step'],3), 'ms  median', round(r.get('ms_per_step_median',0),3), r['roofline']['launch_ms_by_position_in_step'])"
  done
done
done

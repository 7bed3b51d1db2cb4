#!/bin/bash
# small-graph latency: split launches (one workgroup per output segment and 16-edge tile) with 4 vs 8 waves per workgroup, same call
# (variant libraries from tools/build_variants.sh base: nw8:"-DIS_NW=8"; the planner follows with HG_IS_WAVES)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r05sw}; mkdir -p $out
V=$PWD/hamgnn_amd/lib/variants
for rep in 1 2; do
  for n in base:4 nw8:8; do
    name=${n%%:*}; w=${n##*:}
    for wl in si2 si64; do
      HG_IS_WAVES=$w HG_LIB_PATH=$V/lib_$name.so timeout 200 python bench.py --steps 50 --warmup 5 --workload $wl --no-cpu-baseline --no-accuracy --no-mfma-probe 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '$wl', round(r['ms_per_step'],3), round(r.get('ms_per_step_median',0),3), r['roofline']['launch_ms_by_position_in_step'])" >> $out/ab.log
    done
  done
done
cat $out/ab.log
HG_IS_WAVES=8 HG_LIB_PATH=$V/lib_nw8.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "si2 or default_irreps or fixture or golden" 2>&1 | tail -3

#!/bin/bash
# same-call A/B of the two exact shortcuts on the benchmark crystal: complete programs / structural zeros only / both (the default)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r05ab}; mkdir -p $out
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy --no-mfma-probe"
HG_STRUCT_ZEROS=0 HG_DEAD_OUT=0 $B > $out/bench_complete_programs.json 2>/dev/null
HG_DEAD_OUT=0 $B > $out/bench_zeros_only.json 2>/dev/null
$B > $out/bench_default.json 2>/dev/null
python - <<PY
import json
for n in ("complete_programs", "zeros_only", "default"):
    r = json.loads(open("$out/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, round(r["value"]), round(r["ms_per_step"], 2), r["roofline"]["launch_ms_by_position_in_step"], round(r["roofline"]["frac"], 4), r["roofline"]["frac_without_structural_zero_flops"], r["exact_shortcuts"])
PY

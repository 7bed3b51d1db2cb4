#!/bin/bash
# r6: same-library A/B of the radial scale on the half-precision matrix pipe with split operands (default) vs its fp32 form (HG_S_SPLIT=0):
# tests/bench_tp.py (131 072 edges, set-A and set-B, node-fed launch) three rounds each, then bench.py on the 10 k-atom crystal both ways.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06c}; mkdir -p $out
rm -f $out/bench_tp.log
for rep in 1 2 3; do
  for irr in A B; do
    for s in 1 0; do
      HG_S_SPLIT=$s timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --irreps $irr --tag "set$irr-split$s" 2>&1 | tail -1 >> $out/bench_tp.log
    done
  done
done
python - <<PY
import json, collections
d = collections.defaultdict(list)
for l in open("$out/bench_tp.log"):
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    d[r["tag"]].append((r["ms"], r["checksum"]))
for k, v in sorted(d.items()): print(k, " ".join(f"{m:.3f}" for m, _ in v), "checksum", v[0][1])
PY
for s in 1 0; do
  HG_S_SPLIT=$s timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_split$s.json 2> $out/bench_split$s.err
  python - <<PY
import json
r = json.loads(open("$out/bench_split$s.json").read().strip().splitlines()[-1])
print("split=$s", "value", round(r["value"]), "ms", round(r["ms_per_step"], 2), "frac", round(r["roofline"]["frac"], 4), "complete", round(r.get("value_complete_programs", 0)), "by position", r["roofline"]["launch_ms_by_position_in_step"], "accuracy", r.get("accuracy", {}).get("rel_max"))
PY
done

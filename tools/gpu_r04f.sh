#!/bin/bash
# planner-only sweep: per-item overhead allowance of the LPT cost model (HG_ITEM_OVH, MFMA slots)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04f; mkdir -p $out; rm -f $out/bench.log
for rep in 1 2; do
for ovh in 60 0 120 200 300 450; do
  HG_ITEM_OVH=$ovh timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag ovh$ovh 2>&1 | tail -1 >> $out/bench.log
done; done
python - <<PY
import json, collections
d = collections.defaultdict(list)
for l in open("$out/bench.log"):
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    d[r["tag"]].append(r["ms"])
for k, v in d.items(): print(k, " ".join(f"{m:.3f}" for m in v))
PY

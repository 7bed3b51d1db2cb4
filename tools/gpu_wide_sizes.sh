#!/bin/bash
# r5: does the wide schedule win at launch sizes where hg_tp_is underfills the chip?  bench_tp at several edge counts, hg_tp_is (its own choice of parts) vs wide (forced)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-widesz}; mkdir -p $out; rm -f $out/bench.log
for E in 1024 2048 4096 8192 16384 32768 65536; do
  for rep in 1 2; do
    HG_MP_WIDE=0 timeout 60 python tests/bench_tp.py --edges $E --nodes $((E / 8)) --reps 20 --tag is_$E 2>&1 | tail -1 >> $out/bench.log
    HG_MP_WIDE=force timeout 60 python tests/bench_tp.py --edges $E --nodes $((E / 8)) --reps 20 --tag wide_$E 2>&1 | tail -1 >> $out/bench.log
  done
done
python - <<PY
import json, collections
d = collections.defaultdict(list)
for l in open("$out/bench.log"):
    try: r = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    d[r["tag"]].append((r["ms"], r["kernel"]))
for k, v in d.items(): print(k, " ".join(f"{m:.4f}" for m, _ in v), v[0][1])
PY

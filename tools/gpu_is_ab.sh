#!/bin/bash
# same-call A/B of tp_is variant libraries on bench_tp (131 072 edges, set-A, node-fed):  tools/gpu_is_ab.sh <tag> "<variants>"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-isab}; mkdir -p $out
V=hamgnn_amd/lib/variants
rm -f $out/bench.log
for rep in 1 2 3; do
  for n in $2; do
    HG_LIB_PATH=$PWD/$V/lib_$n.so timeout 60 python tests/bench_tp.py --nodes 16384 --reps 8 --tag $n 2>&1 | tail -1 >> $out/bench.log
  done
done
python - <<PY
import json, collections
d = collections.defaultdict(list)
for l in open("$out/bench.log"):
    try: r = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    d[r["tag"]].append((r["ms"], r["checksum"]))
for k, v in d.items(): print(k, " ".join(f"{m:.3f}" for m, _ in v), "checksum", v[0][1])
PY

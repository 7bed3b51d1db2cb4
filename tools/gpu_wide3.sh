#!/bin/bash
# r5: same-call A/B of variant libraries on bench_tp, each with its own environment:  tools/gpu_wide3.sh <tag> "lib:ENV=..,ENV=.. lib2:.."
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-wide3}; mkdir -p $out
V=hamgnn_amd/lib/variants
rm -f $out/bench.log
for rep in 1 2; do
  HG_MP_WIDE=0 timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag is 2>&1 | tail -1 >> $out/bench.log
  for spec in $2; do
    n="${spec%%:*}"; e="${spec#*:}"; [ "$e" = "$spec" ] && e="HG_X=0"
    env ${e//,/ } HG_MP_WIDE=1 HG_LIB_PATH=$PWD/$V/lib_$n.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag "$spec" 2>&1 | tail -1 >> $out/bench.log
  done
done
python - <<PY
import json, collections
d = collections.defaultdict(list)
for l in open("$out/bench.log"):
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    d[r["tag"]].append((r["ms"], r["checksum"]))
for k, v in d.items(): print(k, " ".join(f"{m:.3f}" for m, _ in v), "checksum", v[0][1])
PY

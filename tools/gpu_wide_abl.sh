#!/bin/bash
# r5: timing attribution of csrc/tp_wide.hip by ablation builds (wrong results by construction) + the light phase profile
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-wideabl}; mkdir -p $out
V=hamgnn_amd/lib/variants
rm -f $out/prof.log $out/bench.log
HG_PROF=1 HG_MP_WIDE=1 HG_LIB_PATH=$PWD/$V/lib_profl.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 4 --tag profl 2>&1 | tail -2 >> $out/prof.log
cat $out/prof.log
for rep in 1 2; do
  for n in ${HG_ABL:-nw16 noepi nostage nos nocomp nosc noall}; do
    HG_MP_WIDE=1 HG_LIB_PATH=$PWD/$V/lib_$n.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag $n 2>&1 | tail -1 >> $out/bench.log
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

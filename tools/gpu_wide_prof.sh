#!/bin/bash
# r5: phase profile (HG_PROF build of csrc/tp_wide.hip) + planner knobs of the wide schedule, bench_tp (131 072 edges, set-A, node-fed)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-wideprof}; mkdir -p $out
V=hamgnn_amd/lib/variants
rm -f $out/prof.log $out/bench.log
for tpw in ${HG_TPWS:-1.5 2.5 4}; do
  HG_PROF=1 HG_MP_WIDE=1 HG_WIDE_TPW=$tpw HG_LIB_PATH=$PWD/$V/lib_${HG_PROFLIB:-prof}.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 4 --tag prof_tpw$tpw 2>&1 | tail -2 >> $out/prof.log
done
cat $out/prof.log
for rep in 1 2; do
  HG_MP_WIDE=0 timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag is 2>&1 | tail -1 >> $out/bench.log
  for tpw in ${HG_TPWS:-1.5 2.5 4}; do
    HG_MP_WIDE=1 HG_WIDE_TPW=$tpw HG_LIB_PATH=$PWD/$V/lib_nw16.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag wide_tpw$tpw 2>&1 | tail -1 >> $out/bench.log
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

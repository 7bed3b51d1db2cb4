#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06q}; mkdir -p $out
V=$PWD/hamgnn_amd/lib/variants
for n in z z1 z2 z3; do TAG=$n HG_LIB_PATH=$V/lib_$n.so python tools/gpu_debug_det.py > $out/det_$n.log 2>&1; grep -h tag $out/det_$n.log | cut -c1-300; done
rm -f $out/bench.log
for rep in 1 2; do
  for n in z z1 z2 z3; do
    HG_LIB_PATH=$V/lib_$n.so timeout 60 python tests/bench_tp.py --nodes 16384 --reps 8 --tag $n 2>&1 | tail -1 >> $out/bench.log
  done
  timeout 60 python tests/bench_tp.py --nodes 16384 --reps 8 --tag base 2>&1 | tail -1 >> $out/bench.log
  HG_S_SPLIT=0 timeout 60 python tests/bench_tp.py --nodes 16384 --reps 8 --tag off 2>&1 | tail -1 >> $out/bench.log
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

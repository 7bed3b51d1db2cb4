#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-abl}; mkdir -p $out; rm -f $out/bench.log
V=hamgnn_amd/lib/variants
for rep in 1 2; do for n in $(cat $V/list.txt); do
  HG_LIB_PATH=$PWD/$V/lib_$n.so timeout 120 python tests/bench_tp.py --lite --nodes 16384 --reps 8 --tag $n 2>&1 | tail -1 >> $out/bench.log
done; done
HG_BENCH_LDS=163840 HG_LIB_PATH=$PWD/$V/lib_base.so timeout 120 python tests/bench_tp.py --lite --nodes 16384 --reps 8 --tag base1wg 2>&1 | tail -1 >> $out/bench.log
HG_PROF=1 HG_LIB_PATH=$PWD/$V/lib_prof.so timeout 120 python tests/bench_tp.py --lite --nodes 16384 --reps 4 --tag prof 2>&1 | tail -2 > $out/prof.log
python - <<PY
import json, collections
d = collections.defaultdict(list)
for l in open("$out/bench.log"):
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    d[r["tag"]].append((r["ms"], r["checksum"]))
for k, v in d.items(): print(k, " ".join(f"{m:.3f}" for m, _ in v), "checksum", v[0][1])
print(open("$out/prof.log").read()[:1500])
PY

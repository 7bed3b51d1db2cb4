#!/bin/bash
# r5: the "own" form of the wide schedule: parity tests under HG_WIDE_SCHED=own, then same-call A/B   tools/gpu_wide4.sh <tag> "lib:ENV=..,ENV=.. ..."
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-wide4}; mkdir -p $out
V=hamgnn_amd/lib/variants
rm -f $out/bench.log
HG_WIDE_SCHED=own timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide" > $out/tests_own.log 2>&1; tail -4 $out/tests_own.log
for rep in 1 2; do
  HG_MP_WIDE=0 timeout 60 python tests/bench_tp.py --nodes 16384 --reps 8 --tag is 2>&1 | tail -1 >> $out/bench.log
  for spec in $2; do
    n="${spec%%:*}"; e="${spec#*:}"; [ "$e" = "$spec" ] && e="HG_X=0"
    env ${e//,/ } HG_MP_WIDE=1 HG_LIB_PATH=$PWD/$V/lib_$n.so timeout 60 python tests/bench_tp.py --nodes 16384 --reps 8 --tag "$spec" 2>&1 | tail -1 >> $out/bench.log
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

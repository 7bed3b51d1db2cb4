#!/bin/bash
# r5: wide-schedule iteration: parity tests of the wide kernel, same-call A/B of variant libraries x planner knobs on bench_tp, light phase profile
#   tools/gpu_wide2.sh <tag> "<lib variants>" "<env settings separated by ;>"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-wide2}; mkdir -p $out
V=hamgnn_amd/lib/variants
libs=${2:-nw16}; IFS=';' read -ra envs <<< "${3:-HG_X=0}"
rm -f $out/bench.log $out/prof.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide" > $out/tests_wide.log 2>&1; tail -3 $out/tests_wide.log
if [ -f $V/lib_profl.so ]; then
  for e in "${envs[@]}"; do
    env $e HG_PROF=1 HG_MP_WIDE=1 HG_LIB_PATH=$PWD/$V/lib_profl.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 4 --tag "profl $e" 2>&1 | tail -2 >> $out/prof.log
  done
  cat $out/prof.log
fi
for rep in 1 2; do
  HG_MP_WIDE=0 timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag is 2>&1 | tail -1 >> $out/bench.log
  for n in $libs; do
    for e in "${envs[@]}"; do
      env $e HG_MP_WIDE=1 HG_LIB_PATH=$PWD/$V/lib_$n.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag "$n $e" 2>&1 | tail -1 >> $out/bench.log
    done
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

#!/bin/bash
# r5: same-call A/B of variant libraries of the wide kernel (csrc/tp_wide.hip) against hg_tp_is on bench_tp (131 072 edges, set-A, node-fed), each with its own
# environment (planner knobs):      tools/gpu_wide_ab.sh <tag> "lib:ENV=..,ENV=.. lib2:.. ..." [tests]
#   libraries: HG_VARIANT_FILES=tp_wide tools/build_variants.sh nw16: nw12:"-DWD_NW=12" a2e0:"-DWD_A2_EARLY=0" ...   (the ablation hooks -DWD_ABL_* behind the table in
#   profiles/r05_tp_wide.md were removed from the source once the question was answered: commits c8418b7..7e1e420 have them)
#   environment: HG_WIDE_SCHED=pools|own, HG_WIDE_TPW, HG_WIDE_COST_REC, HG_WIDE_COST_STAGE, HG_WIDE_STAGE_POS, HG_WIDE_WAVES (must match -DWD_NW)
#   a third argument `tests` runs the wide parity tests first (under the first spec's HG_WIDE_SCHED)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-wideab}; mkdir -p $out
V=hamgnn_amd/lib/variants
rm -f $out/bench.log
if [ "${3:-}" = "tests" ]; then
  timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide" > $out/tests_wide.log 2>&1; tail -4 $out/tests_wide.log
  HG_WIDE_SCHED=own timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide" > $out/tests_wide_own.log 2>&1; tail -4 $out/tests_wide_own.log
fi
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

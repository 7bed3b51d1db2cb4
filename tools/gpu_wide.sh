#!/bin/bash
# r5: the wide schedule (csrc/tp_wide.hip) on the GPU box: its parity tests, then same-call A/B against hg_tp_is on bench_tp (131 072 edges, set-A,
# node-fed) for the workgroup sizes built as variants (hamgnn_amd/lib/variants/lib_nw*.so, tools/build_variants.sh), then the whole forward.
#   tools/gpu_wide.sh <tag> [tests|ab|bench ...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-wide}; mkdir -p $out; shift
what="${@:-tests ab bench}"
V=hamgnn_amd/lib/variants
if [[ " $what " == *" tests "* ]]; then
  timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide" > $out/tests_wide.log 2>&1; tail -15 $out/tests_wide.log
fi
if [[ " $what " == *" ab "* ]]; then
  rm -f $out/bench.log
  for rep in 1 2; do
    HG_MP_WIDE=0 timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag is 2>&1 | tail -1 >> $out/bench.log
    for nw in 16 12 8; do
      [ -f $V/lib_nw$nw.so ] || continue
      for tpw in ${HG_TPWS:-2.5}; do
        HG_MP_WIDE=1 HG_WIDE_WAVES=$nw HG_WIDE_TPW=$tpw HG_LIB_PATH=$PWD/$V/lib_nw$nw.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 8 --tag wide_nw${nw}_tpw$tpw 2>&1 | tail -1 >> $out/bench.log
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
fi
if [[ " $what " == *" bench "* ]]; then
  HG_MP_WIDE=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy > $out/bench_is.json 2> $out/bench_is.err; tail -c 1500 $out/bench_is.json
  HG_MP_WIDE=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_wide.json 2> $out/bench_wide.err; tail -c 1500 $out/bench_wide.json
fi

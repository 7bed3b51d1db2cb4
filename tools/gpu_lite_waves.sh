#!/bin/bash
# lite_mode with 8 waves per workgroup (four per SIMD) vs 4: parity tests, launch timing, bench line.  Needs variants nw4 (-DIS_NW_LITE=4) and nw8.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-lw}; mkdir -p $out; rm -f $out/bench.log
V=$PWD/hamgnn_amd/lib/variants
timeout 900 python -m pytest tests -x -q -m gpu -k "lite" > $out/tests.log 2>&1; tail -3 $out/tests.log
for rep in 1 2 3; do
  HG_LITE_WAVES=4 HG_LIB_PATH=$V/lib_nw4.so timeout 120 python tests/bench_tp.py --lite --nodes 16384 --reps 8 --tag nw4 2>&1 | tail -1 | cut -c1-200 >> $out/bench.log
  HG_LITE_WAVES=8 HG_LIB_PATH=$V/lib_nw8.so timeout 120 python tests/bench_tp.py --lite --nodes 16384 --reps 8 --tag nw8 2>&1 | tail -1 | cut -c1-200 >> $out/bench.log
done
cat $out/bench.log | python -c "
import sys, json, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    try: r = json.loads(l[:l.rindex(',')] + '}') if not l.strip().endswith('}') else json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    d[r['tag']].append(r['ms'])
for k, v in d.items(): print(k, ' '.join(f'{m:.3f}' for m in v))
"
if [ "${2:-}" = bench ]; then python bench.py --lite --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_lite.json 2> $out/bench.err; cut -c1-300 $out/bench_lite.json; tail -3 $out/bench.err; fi

#!/bin/bash
# lite_mode streams (plan._lite_streams / stream_lite) vs the r3/r4 runs (HG_LITE_STREAMS=0): parity tests + launch timing (+ bench line with "bench")
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-lst}; mkdir -p $out; rm -f $out/bench.log
timeout 900 python -m pytest tests -x -q -m gpu -k "lite" > $out/tests.log 2>&1; tail -3 $out/tests.log
for rep in 1 2 3; do for s in 1; do
  HG_LITE_STREAMS=$s timeout 120 python tests/bench_tp.py --lite --nodes 16384 --reps 8 --tag streams$s 2>&1 | tail -1 | cut -c1-220 >> $out/bench.log
done; done
cat $out/bench.log | python -c "
import sys, json, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    d[r['tag']].append((r['ms'], r['checksum']))
for k, v in d.items(): print(k, ' '.join(f'{m:.3f}' for m, _ in v), 'checksum', v[0][1])
"
if [ "${2:-}" = bench ]; then python bench.py --lite --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_lite.json 2> $out/bench.err; cut -c1-300 $out/bench_lite.json; fi

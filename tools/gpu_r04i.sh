#!/bin/bash
# lite_mode with paired steps (HG_LITE_PAIR=0: r3 streams): parity + launch timing + bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04i; mkdir -p $out; rm -f $out/bench.log
timeout 900 python -m pytest tests -m gpu -x -q -k "lite" > $out/tests.log 2>&1; tail -3 $out/tests.log
for p in 1 0 1 0; do HG_LITE_PAIR=$p timeout 120 python tests/bench_tp.py --lite --nodes 16384 --reps 8 --tag pair$p 2>&1 | tail -1 | cut -c1-200 >> $out/bench.log; done
cat $out/bench.log
for p in 1 0; do
HG_LITE_PAIR=$p python bench.py --lite --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_lite_pair$p.json 2>> $out/bench.err
python -c "
import json; d = json.loads(open('$out/bench_lite_pair$p.json').read().strip().splitlines()[-1]); print('pair=$p', round(d['value']), round(d['ms_per_step'], 3), d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('accuracy'))"
done

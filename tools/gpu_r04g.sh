#!/bin/bash
# fused node scatter: parity + bench A/B (HG_FUSED_SCATTER=0 restores message rows + hg_segment_sum)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04g; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "fused_node_scatter or sio2_setA or si2_default or full_size or sharded_forward or backbone_golden or training_step_is_bit" > $out/tests.log 2>&1; tail -4 $out/tests.log
for f in 1 0; do
  HG_FUSED_SCATTER=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy > $out/bench_fused$f.json 2>> $out/bench.err
  python -c "
import json; d = json.loads(open('$out/bench_fused$f.json').read().strip().splitlines()[-1]); print('fused=$f', round(d['value']), round(d['ms_per_step'], 3), d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03ad; mkdir -p $out
for so in default hamgnn_amd/lib/variants/lib_ring8.so hamgnn_amd/lib/variants/lib_ring2.so; do
  if [ $so = default ]; then python bench.py --lite --steps 6 --warmup 2 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  else HG_LIB_PATH=$so python bench.py --lite --steps 6 --warmup 2 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', d['ms_per_step'], d['roofline']['avg_launch_ms'])"; fi
done
HG_LITE_FOLD=0 python bench.py --lite --steps 6 --warmup 2 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfolded', d['ms_per_step'], d['roofline']['avg_launch_ms'])"

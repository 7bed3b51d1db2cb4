#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for so in hamgnn_amd/lib/variants/lib_*.so; do
HG_LIB_PATH=$so python bench.py --lite --steps 6 --warmup 2 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done

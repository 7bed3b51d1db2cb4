#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HG_LIB_PATH=hamgnn_amd/lib/variants/lib_noitems.so python bench.py --lite --steps 6 --warmup 2 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lite noitems', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
HG_LIB_PATH=hamgnn_amd/lib/variants/lib_noitems.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full noitems', d['ms_per_step'], d['roofline']['avg_launch_ms'])"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03ah; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sio2 or fixture or oracle or lite or doping" > $out/tests.log 2>&1; tail -2 $out/tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('emb auto', d['value'], d['ms_per_step'])"
HG_EMB_KERNEL=seg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('emb seg', d['value'], d['ms_per_step'])"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03w; mkdir -p $out
for i in 1 2 3; do python tests/bench_tp.py --nodes 16384 --tag fdiv 2>&1 | grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print(d['tag'], d['kernel'], round(d['ms'],3), d['checksum'])"; done
python tests/bench_tp.py --nodes 16384 --irreps B 2>&1 | grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print(d['tag'], d['kernel'], round(d['ms'],3), d['checksum'])"
python tests/bench_tp.py --nodes 16384 --adjoint 2>&1 | grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('adjoint', d['kernel'], round(d['ms'],3), d['checksum'])"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "message_pack or sio2 or random or fixture" > $out/tests.log 2>&1; tail -2 $out/tests.log

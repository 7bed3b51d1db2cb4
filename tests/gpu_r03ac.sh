#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03ac; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lite" > $out/tests.log 2>&1; tail -3 $out/tests.log
HG_PROF=1 python tests/bench_tp.py --nodes 16384 2>&1 | grep '^{' | head -3 | cut -c1-300

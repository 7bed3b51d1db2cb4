#!/bin/bash
# run a -k selection of the -m gpu suite:  tools/gpu_tests_k.sh <outdir> "<-k expression>"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-tk}; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "$2" > $out/tests.log 2>&1; tail -15 $out/tests.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-profl}; mkdir -p $out
HG_PROF=1 HG_MP_WIDE=1 HG_LIB_PATH=$PWD/hamgnn_amd/lib/variants/lib_profl.so timeout 60 python tests/bench_tp.py --nodes 16384 --reps 4 --tag profl 2>&1 | tail -2 | tee $out/prof.log

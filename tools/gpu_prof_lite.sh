#!/bin/bash
# per-phase shader-clock profile (-DHG_PROF build in hamgnn_amd/lib/variants/lib_prof.so) of one lite and one default launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-prof}; mkdir -p $out
V=$PWD/hamgnn_amd/lib/variants
HG_PROF=1 HG_LIB_PATH=$V/lib_prof.so timeout 120 python tests/bench_tp.py --lite --nodes 16384 --reps 4 --tag prof 2>&1 | tail -1 > $out/prof_lite.json
HG_PROF=1 HG_LIB_PATH=$V/lib_prof.so timeout 120 python tests/bench_tp.py --nodes 16384 --reps 4 --tag prof 2>&1 | tail -1 > $out/prof_default.json
cat $out/prof_lite.json $out/prof_default.json
[ -n "$2" ] && timeout 900 python -m pytest tests -x -q -m gpu -k "$2" 2>&1 | tail -3

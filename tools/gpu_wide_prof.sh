#!/bin/bash
# r5: phase profile of the wide kernel: HG_VARIANT_FILES=tp_wide tools/build_variants.sh prof:"-DHG_PROF" profl:"-DHG_PROF -DHG_PROF_LITE"
# (prof: probes around every record stage -- distorts; profl: only around the pool barriers, the zero fill and the epilogue.  The launch time of a profile build is
#  inflated by the final atomics of 131 072 waves on 16 counters; the FRACTIONS are taken before them)      tools/gpu_wide_prof.sh <tag> [prof|profl]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-wideprof}; mkdir -p $out
HG_PROF=1 HG_MP_WIDE=1 HG_LIB_PATH=$PWD/hamgnn_amd/lib/variants/lib_${2:-profl}.so timeout 60 python tests/bench_tp.py --nodes 16384 --reps 4 --tag ${2:-profl} 2>&1 | tail -2 | tee $out/prof.log

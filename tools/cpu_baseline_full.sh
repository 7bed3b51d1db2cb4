#!/bin/bash
# r6: the CPU port on a WHOLE BASELINE configuration -- si512, set-B, 43 k edges -- with a thread sweep up to the host's count, 3 forwards each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06cpu}; mkdir -p $out
HIP_VISIBLE_DEVICES="" timeout 1500 python bench.py --cpu-baseline-only --cpu-baseline-full --workload si512 --irreps B > $out/cpu_full_si512_setB.log 2>&1
tail -2 $out/cpu_full_si512_setB.log
HIP_VISIBLE_DEVICES="" HG_CPU_THREADS=${2:-32} timeout 1500 python bench.py --cpu-baseline-only --cpu-baseline-full --workload sio2_300 --irreps A > $out/cpu_full_sio2_300_setA.log 2>&1
tail -2 $out/cpu_full_sio2_300_setA.log

#!/bin/bash
# r6: the CPU port on a WHOLE BASELINE configuration -- si512, set-B, 44 k edges -- thread sweep 8 / 16 / 32 / 64, 3 forwards per count (about ten minutes of host time)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06cpu}; mkdir -p $out
HIP_VISIBLE_DEVICES="" timeout 1200 python bench.py --cpu-baseline-only --cpu-baseline-full --workload si512 --irreps B > $out/cpu_full_si512_setB.log 2>&1
tail -2 $out/cpu_full_si512_setB.log

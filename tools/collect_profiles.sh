#!/bin/bash
# Round-end evidence, run ON THE GPU BOX (gpurun): bench lines, rocprofv3 kernel stats of the benchmarked command, PMC passes of the
# dominant kernel (separate --pmc passes; never combined with trace domains other than the kernel trace).  Output: gpurun_out/$1/
set -u
tag=${1:-r02_final}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench_sio2_10k_setA.json 2> $out/bench_sio2.err
python bench.py --steps 20 --warmup 5 --workload si512 --irreps B > $out/bench_si512_setB.json 2>> $out/bench_sio2.err
python bench.py --steps 20 --warmup 5 --workload mos2_1200 --no-cpu-baseline > $out/bench_mos2_1200_setA.json 2>> $out/bench_sio2.err
python bench.py --steps 50 --warmup 5 --workload si2 --no-cpu-baseline > $out/bench_si2_setA.json 2>> $out/bench_sio2.err
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_profiled.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/sio2_10k_kernel_stats.csv
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
head -1 $f > $out/sio2_10k_kernel_trace_tp_is.csv; grep "tp_is_kernel" $f >> $out/sio2_10k_kernel_trace_tp_is.csv
rm -rf $out/prof
# PMC of one node-fed MessagePackBlock launch (tests/bench_tp.py), one counter group per pass
pmc() { rocprofv3 --pmc $2 -d $out/pmc_$1 --output-format csv -- timeout 150 python tests/bench_tp.py --reps 2 --nodes 16384 --irreps ${3:-A} > $out/pmc_$1.log 2>&1;
        cp $(find $out/pmc_$1 -name "*counter_collection.csv" | head -1) $out/pmc_$1.csv 2>/dev/null; rm -rf $out/pmc_$1; }
pmc sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVES"
pmc sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
pmc fetchB "FETCH_SIZE" B
pmc writeB "WRITE_SIZE" B
ls -la $out

#!/bin/bash
# Round-6 evidence, run ON THE GPU BOX (gpurun): bench line of the default command, rocprofv3 kernel stats of the benchmarked command, PMC passes ON THE
# BENCHMARKED LAUNCHES (separate --pmc passes; never combined with trace domains other than the kernel trace): pipe occupancy and HBM bytes of the
# fused-scatter (ConvBlock) and per-edge-output (PairInteractionBlock) launches of tp_is_kernel, the kernel list of the 2-atom cell.  Output: gpurun_out/$1/
set -u
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench_sio2_10k_setA.json 2> $out/bench_sio2.err
if [ "${2:-}" = "full" ]; then
python bench.py --steps 20 --warmup 5 --workload si512 --irreps B --no-accuracy > $out/bench_si512_setB.json 2>> $out/bench_sio2.err
python bench.py --steps 20 --warmup 5 --workload mos2_1200 --no-cpu-baseline --no-accuracy > $out/bench_mos2_1200_setA.json 2>> $out/bench_sio2.err
python bench.py --steps 20 --warmup 5 --workload mos2_1200 --soc > $out/bench_mos2_1200_setA_soc.json 2>> $out/bench_sio2.err
python bench.py --steps 50 --warmup 5 --workload si2 --no-cpu-baseline --no-accuracy > $out/bench_si2_setA.json 2>> $out/bench_sio2.err
python bench.py --workload uni8 --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy > $out/bench_uni8_setA.json 2>> $out/bench_sio2.err
python bench.py --lite --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sio2_10k_setA_lite.json 2>> $out/bench_sio2.err
fi
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-accuracy --no-mfma-probe"
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- $B > $out/bench_profiled.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/sio2_10k_kernel_stats.csv
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
head -1 $f > $out/sio2_10k_kernel_trace_tp_is.csv; grep "tp_is_kernel" $f >> $out/sio2_10k_kernel_trace_tp_is.csv
rm -rf $out/prof
rocprofv3 --kernel-trace --stats -d $out/prof2 --output-format csv -- python bench.py --steps 5 --warmup 2 --workload si2 --no-cpu-baseline --no-accuracy --no-mfma-probe > $out/bench_si2_profiled.json 2>> $out/prof.err
cp $(find $out/prof2 -name "*kernel_stats.csv" | head -1) $out/si2_kernel_stats.csv
rm -rf $out/prof2
pmc() { rocprofv3 --pmc $2 -d $out/pmc_$1 --output-format csv -- timeout 200 $3 > $out/pmc_$1.log 2>&1;
        cp $(find $out/pmc_$1 -name "*counter_collection.csv" | head -1) $out/pmc_$1.csv 2>/dev/null; rm -rf $out/pmc_$1; }
pmc sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVES" "$B"
pmc fetch "FETCH_SIZE" "$B"
pmc write "WRITE_SIZE" "$B"
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/pmc_*.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "tp_is_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    kind = {d: ("conv (fused scatter)" if k % 2 == 0 else "pair (one row per edge)") for k, d in enumerate(ids)}     # launches alternate ConvBlock / PairInteractionBlock
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        a = acc[(kind[int(r["Dispatch_Id"])], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f.split("/")[-1], len(ids), "tp_is dispatches")
    for k in sorted(acc): print("   ", k, acc[k][0] / acc[k][1])
PY
ls -la $out

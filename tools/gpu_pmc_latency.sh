#!/bin/bash
# PMC passes that measure latencies and stall reasons of tp_is_kernel (default and lite launch): SQ_INST_LEVEL_* / SQ_INSTS_* = mean latency per
# instruction class, scalar / instruction cache hit rates, active cycles by class.  Run ON THE GPU BOX; output gpurun_out/$1/pmc_<kind>_<pass>.csv
set -u
tag=${1:-pmclat}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
pmc() { rocprofv3 --pmc $2 -d $out/pmc_$1 --output-format csv -- timeout 150 $3 > $out/pmc_$1.log 2>&1;
        cp $(find $out/pmc_$1 -name "*counter_collection.csv" | head -1) $out/pmc_$1.csv 2>/dev/null; rm -rf $out/pmc_$1; }
for kind in ${2:-def lite}; do
  TP="python tests/bench_tp.py --reps 2 --nodes 16384 --irreps A"
  [ $kind = lite ] && TP="$TP --lite"
  pmc ${kind}_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "$TP"
  pmc ${kind}_b "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_INSTS_VALU" "$TP"
  pmc ${kind}_c "SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_DCACHE_HITS SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL" "$TP"
  pmc ${kind}_d "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM" "$TP"
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/pmc_*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "tp_is_kernel" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f.split("/")[-1], {k: v[0] / max(v[1], 1) for k, v in acc.items()})
PY

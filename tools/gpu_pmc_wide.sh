#!/bin/bash
# r5: PMC passes of the wide kernel (csrc/tp_wide.hip) on bench_tp (131 072 edges, set-A, node-fed): pipe occupancy, stall reasons, instruction mix,
# instruction / scalar cache hit rates.  Run ON THE GPU BOX; output gpurun_out/$1/pmc_<pass>.csv.   tools/gpu_pmc_wide.sh <tag> [lib variant] [passes]
set -u
tag=${1:-pmcwide}; libv=${2:-}; passes=${3:-a b c d}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
[ -n "$libv" ] && export HG_LIB_PATH=$PWD/hamgnn_amd/lib/variants/lib_$libv.so
export HG_MP_WIDE=${HG_MP_WIDE:-1}
pmc() { rocprofv3 --pmc $2 -d $out/pmc_$1 --output-format csv -- timeout 150 $3 > $out/pmc_$1.log 2>&1;
        cp $(find $out/pmc_$1 -name "*counter_collection.csv" | head -1) $out/pmc_$1.csv 2>/dev/null; rm -rf $out/pmc_$1; }
TP="python tests/bench_tp.py --reps 2 --nodes 16384 --irreps A"
for p in $passes; do
  case $p in
    a) pmc a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "$TP";;
    b) pmc b "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$TP";;
    c) pmc c "SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_DCACHE_HITS SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL" "$TP";;
    d) pmc d "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM" "$TP";;
    f) pmc f "FETCH_SIZE" "$TP";;
    w) pmc w "WRITE_SIZE" "$TP";;
    t) pmc t "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$TP";;
  esac
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/pmc_*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "tp_wide_kernel" not in r["Kernel_Name"] and "tp_is_kernel" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f.split("/")[-1], {k: v[0] / max(v[1], 1) for k, v in acc.items()})
PY

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03ak; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|INST_CACHE|SQC_" | head -40 > $out/counters.txt; head -40 $out/counters.txt
pmc() { rocprofv3 --pmc $2 -d $out/pmc_$1 --output-format csv -- timeout 150 python tests/bench_tp.py --reps 2 --nodes 16384 --irreps A > $out/pmc_$1.log 2>&1;
        cp $(find $out/pmc_$1 -name "*counter_collection.csv" | head -1) $out/pmc_$1.csv 2>/dev/null; rm -rf $out/pmc_$1; }
pmc ic "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES"
python - <<'PY'
import csv, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open('gpurun_out/r03ak/pmc_ic.csv')):
        if 'tp_is_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    disp = max(1, len({1}))
    for k, v in agg.items(): print(k, v)
except Exception as e:
    print("no pmc csv", e); print(open('gpurun_out/r03ak/pmc_ic.log').read()[-800:])
PY

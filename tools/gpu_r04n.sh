#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04n; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python tests/bench_training.py --workload si512 --steps 6 > $out/train.log 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/training_si512_kernel_stats.csv; rm -rf $out/prof
grep "^step" $out/train.log | tail -2
python - <<PY
import csv
rows = list(csv.DictReader(open("$out/training_si512_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total GPU ms (6 steps incl. first two compile steps)", tot/1e6)
for r in rows[:28]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms", f"{float(r['Percentage']):6.2f} %")
PY

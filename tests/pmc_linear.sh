#!/bin/bash
# PMC passes over tests/bench_linear.py (run ON THE GPU BOX): prints per-kernel counter values of the streaming and the program Linear
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_lin; mkdir -p $out
for c in "$@"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $c -d $out/p_$n --output-format csv -- timeout 120 python tests/bench_linear.py --reps 1 > $out/$n.log 2>&1
  f=$(find $out/p_$n -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import sys, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:28]
    if "linear_planar" in k or "tp_fused" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in d.items():
        print(k, c, "calls=%d" % len(v), "lin1=%.4g lin2=%.4g" % (v[3], v[7]) if len(v) >= 8 else v)
PY
  rm -rf $out/p_$n
done

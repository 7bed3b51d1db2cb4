#!/bin/bash
# late r5: launch count / latency of small crystals after the node-level chain moved to a row program + merged linear_up launch and the sparsity ratio is kept per graph
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r05sg}; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixture or golden or default_irreps or si2 or uni or shard or oracle or unread or structural or attribute or front_door or corr" > $out/tests.log 2>&1
echo "tests exit $?"; tail -3 $out/tests.log
for env in "HG_NODE_ROWPROG=0" "HG_NODE_ROWPROG=1"; do
  for wl in si2 si64; do
    env $env timeout 200 python bench.py --steps 100 --warmup 10 --workload $wl --no-cpu-baseline --no-accuracy --no-mfma-probe 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env', '$wl', round(r['value']), 'edges/s', round(r['ms_per_step'],3), 'ms  median', round(r.get('ms_per_step_median',0),3))"
  done
  env $env timeout 300 python bench.py --workload uni8 --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env', 'uni8', round(r['value']), round(r['ms_per_step'],2), {k: v for k, v in r.items() if 'per_crystal' in k or 'one_crystal' in k})"
done
rocprofv3 --kernel-trace --stats -d $out/prof2 --output-format csv -- python bench.py --steps 5 --warmup 2 --workload si2 --no-cpu-baseline --no-accuracy --no-mfma-probe > $out/bench_si2_profiled.json 2> $out/prof.err
cp $(find $out/prof2 -name "*kernel_stats.csv" | head -1) $out/si2_kernel_stats.csv; rm -rf $out/prof2
python - <<PY
import csv
rows = list(csv.DictReader(open("$out/si2_kernel_stats.csv")))
calls = sum(int(r["Calls"]) for r in rows); cp = sum(int(r["Calls"]) for r in rows if "copyBuffer" in r["Name"])
print("si2 profiled: launches", calls, "of which copyBuffer (compile-time uploads mostly)", cp, "-> per forward (7 forwards)", (calls - cp) / 7.0)
PY

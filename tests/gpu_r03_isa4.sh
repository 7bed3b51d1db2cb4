#!/bin/bash
# post-op fragment requests ahead of the radial phase, batched tile read-modify-write of the plain Linear items: parity + timing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03isa4; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lite or message_pack or tp_ or backward_data or pair or conv or full_model" > $out/tests.log 2>&1; tail -2 $out/tests.log
for i in 1 2; do
  timeout 120 python tests/bench_tp.py --nodes 16384 --lite --tag new >> $out/tp.jsonl 2>> $out/err.log
  timeout 120 python tests/bench_tp.py --nodes 16384 --tag new_default >> $out/tp.jsonl 2>> $out/err.log
done
python bench.py --no-cpu-baseline --no-accuracy > $out/bench_default.json 2>> $out/err.log
python bench.py --lite --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy > $out/bench_lite.json 2>> $out/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r03isa4/tp.jsonl"):
    d = json.loads(l); print(d["tag"], d["kernel"], round(d["ms"], 3), round(d["issued_TF"], 1), d["checksum"])
for f in ("default", "lite"):
    d = json.loads(open(f"gpurun_out/r03isa4/bench_{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"]), round(d["ms_per_step"], 2), d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY

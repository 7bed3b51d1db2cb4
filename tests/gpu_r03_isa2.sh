#!/bin/bash
# lite run loop after the ISA audit (ring request order, batched tile read-modify-write): parity + timing vs the r3 build (variants/lib_base.so)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03isa2; mkdir -p $out
V=hamgnn_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lite or message_pack or tp_ or backward_data" > $out/tests.log 2>&1; tail -2 $out/tests.log
for i in 1 2; do
  timeout 120 python tests/bench_tp.py --nodes 16384 --lite --tag new >> $out/lite.jsonl 2>> $out/err.log
  HG_LIB_PATH=$V/lib_base.so timeout 120 python tests/bench_tp.py --nodes 16384 --lite --tag base >> $out/lite.jsonl 2>> $out/err.log
done
timeout 120 python tests/bench_tp.py --nodes 16384 --lite --adjoint --tag new_adj >> $out/lite.jsonl 2>> $out/err.log
HG_LIB_PATH=$V/lib_base.so timeout 120 python tests/bench_tp.py --nodes 16384 --lite --adjoint --tag base_adj >> $out/lite.jsonl 2>> $out/err.log
timeout 120 python tests/bench_tp.py --nodes 16384 --tag new_default >> $out/lite.jsonl 2>> $out/err.log
python bench.py --lite --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sio2_10k_setA_lite.json 2>> $out/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r03isa2/lite.jsonl"):
    d = json.loads(l); print(d["tag"], d["kernel"], round(d["ms"], 3), round(d["issued_TF"], 1), d["checksum"])
d = json.loads(open("gpurun_out/r03isa2/bench_sio2_10k_setA_lite.json").read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"], 2), d.get("accuracy", {}).get("rel_max"))
PY

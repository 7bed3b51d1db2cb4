#!/bin/bash
# A/B of the ISA-audit changes to tp_is.hip (r3, late): RTM = 1 GEMM1 reads batched (sched_group_barrier), lite run ring fixes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03isa; mkdir -p $out
V=hamgnn_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lite or message_pack or tp_ or backward_data" > $out/tests.log 2>&1; tail -2 $out/tests.log
for name in new base touch nosgb new base touch; do
  HG_LIB_PATH=$V/lib_$name.so timeout 120 python tests/bench_tp.py --nodes 16384 --tag $name >> $out/default.jsonl 2>> $out/err.log
done
for name in new base sdesc new base sdesc; do
  HG_LIB_PATH=$V/lib_$name.so timeout 120 python tests/bench_tp.py --nodes 16384 --lite --tag $name >> $out/lite.jsonl 2>> $out/err.log
done
HG_LIB_PATH=$V/lib_new.so timeout 120 python tests/bench_tp.py --nodes 16384 --adjoint --tag new_adj >> $out/default.jsonl 2>> $out/err.log
HG_LIB_PATH=$V/lib_base.so timeout 120 python tests/bench_tp.py --nodes 16384 --adjoint --tag base_adj >> $out/default.jsonl 2>> $out/err.log
python - <<'PY'
import json
for f in ("default", "lite"):
    for l in open(f"gpurun_out/r03isa/{f}.jsonl"):
        d = json.loads(l); print(f, d["tag"], d["kernel"], round(d["ms"], 3), round(d["issued_TF"], 1), d["checksum"])
PY

#!/bin/bash
# epilogue with the four channel groups of a unit unrolled (-DHG_EPI_UNROLL) vs the rolled loop
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03isa5; mkdir -p $out
V=hamgnn_amd/lib/variants
for i in 1 2 3; do
  HG_LIB_PATH=$V/lib_epi.so timeout 120 python tests/bench_tp.py --nodes 16384 --tag epi >> $out/tp.jsonl 2>> $out/err.log
  timeout 120 python tests/bench_tp.py --nodes 16384 --tag base >> $out/tp.jsonl 2>> $out/err.log
done
HG_LIB_PATH=$V/lib_epi.so timeout 120 python tests/bench_tp.py --nodes 16384 --lite --tag epi_lite >> $out/tp.jsonl 2>> $out/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r03isa5/tp.jsonl"):
    d = json.loads(l); print(d["tag"], d["kernel"], round(d["ms"], 3), round(d["issued_TF"], 1), d["checksum"])
PY

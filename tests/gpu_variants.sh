# usage: gpu_variants.sh <outdir> <bench_tp args...> ; runs every variant lib with HG_ST=1 and 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; shift; mkdir -p $out
for so in default hamgnn_amd/lib/variants/lib_*.so; do
  for st in 1 0; do
    if [ $so = default ]; then HG_ST=$st timeout 120 python tests/bench_tp.py "$@" --tag st$st 2>&1 | tail -1
    else HG_ST=$st HG_LIB_PATH=$so timeout 120 python tests/bench_tp.py "$@" --tag st$st 2>&1 | tail -1; fi
  done
done | tee $out/variants.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l); print(d['lib'], d['tag'], d['kernel'], round(d['ms'], 3), d['checksum'])
    except Exception: print(l.strip()[:200])"

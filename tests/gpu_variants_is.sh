# usage: gpu_variants_is.sh <outdir> <bench_tp args...> ; the default library and every variant, input-stationary kernel, two runs each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; shift; mkdir -p $out
for rep in 1 2; do
for so in default hamgnn_amd/lib/variants/lib_*.so; do
  if [ $so = default ]; then timeout 120 python tests/bench_tp.py "$@" --tag run$rep 2>&1 | tail -1
  else HG_LIB_PATH=$so timeout 120 python tests/bench_tp.py "$@" --tag run$rep 2>&1 | tail -1; fi
done; done | tee $out/variants.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l); print(d['lib'], d['tag'], d['kernel'], round(d['ms'], 3), d['checksum'])
    except Exception: print(l.strip()[:200])"

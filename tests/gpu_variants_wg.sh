# usage: gpu_variants_wg.sh <outdir> <bench_wgrad args...> ; the default library and every variant
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; shift; mkdir -p $out
for so in default hamgnn_amd/lib/variants/lib_*.so; do
  if [ $so = default ]; then timeout 120 python tests/bench_wgrad.py "$@" 2>&1 | grep '^{'
  else HG_LIB_PATH=$so timeout 120 python tests/bench_wgrad.py "$@" 2>&1 | grep '^{'; fi
done | tee $out/variants.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l); print(d['lib'], d['nsplit'], round(d['ms'], 3), d['checksum'])
    except Exception: print(l.strip()[:200])"

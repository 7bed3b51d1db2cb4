# usage: gpu_quick.sh <outdir>: streamed-kernel parity subset + one-launch A/B against the input-stationary kernel (set-A, set-B)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "static_stream or sio2_setA" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
for st in 1 0; do HG_ST=$st timeout 200 python tests/bench_tp.py --nodes 16384 --reps 8 --tag A_st$st 2>&1 | tail -1; done > $out/bench_tp.jsonl
for st in 1 0; do HG_ST=$st timeout 200 python tests/bench_tp.py --irreps B --nodes 16384 --reps 8 --tag B_st$st 2>&1 | tail -1; done >> $out/bench_tp.jsonl
tail -3 $out/pytest.log; python -c "
import sys, json
for l in open('$out/bench_tp.jsonl'):
    try: d = json.loads(l); print(d['tag'], d['kernel'], round(d['ms'], 3), d['checksum'])
    except Exception: print(l.strip()[:300])"

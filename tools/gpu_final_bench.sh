#!/bin/bash
# the non-default bench lines of profiles/r03_bench_*.json on the final build (the default line + tests + smoke: tests/gpu_final_short.sh)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-finalb}; mkdir -p $out
python bench.py --steps 20 --warmup 5 --workload si512 --irreps B --no-accuracy > $out/bench_si512_setB.json 2>> $out/bench.err
python bench.py --steps 20 --warmup 5 --workload mos2_1200 --no-cpu-baseline --no-accuracy > $out/bench_mos2_1200_setA.json 2>> $out/bench.err
python bench.py --steps 20 --warmup 5 --workload mos2_1200 --soc --no-cpu-baseline --no-accuracy > $out/bench_mos2_1200_setA_soc.json 2>> $out/bench.err
python bench.py --steps 50 --warmup 5 --workload si2 --no-cpu-baseline --no-accuracy > $out/bench_si2_setA.json 2>> $out/bench.err
python bench.py --workload uni8 --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy > $out/bench_uni8_setA.json 2>> $out/bench.err
python bench.py --lite --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_sio2_10k_setA_lite.json 2>> $out/bench.err
for f in $out/bench_*.json; do python -c "
import json, sys; d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('mfma_probe_tflops'))"; done

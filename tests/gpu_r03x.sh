#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03x; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "head or soc or su2 or sio2 or fixture or oracle or uni or band" > $out/tests.log 2>&1; tail -4 $out/tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy > $out/bench_rowprog.json 2> $out/bench.err; python -c "
import json; d = json.loads(open('$out/bench_rowprog.json').read().strip().splitlines()[-1]); print('rowprog', d['value'], d['ms_per_step'])"
HG_ROWPROG=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy > $out/bench_norowprog.json 2>> $out/bench.err; python -c "
import json; d = json.loads(open('$out/bench_norowprog.json').read().strip().splitlines()[-1]); print('separate', d['value'], d['ms_per_step'])"
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-accuracy > $out/bench_profiled.json 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/sio2_10k_kernel_stats.csv; rm -rf $out/prof
head -12 $out/sio2_10k_kernel_stats.csv | cut -c1-140

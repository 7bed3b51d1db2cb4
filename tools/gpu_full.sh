#!/bin/bash
# the whole -m gpu suite + smoke() + default bench line, as the driver runs them at round end
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-full}; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py > $out/bench_default.json 2> $out/bench.err; python -c "
import json; d = json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('accuracy', {}).get('rel_max'), d.get('cpu_baseline', {}).get('value'))"

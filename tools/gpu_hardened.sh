# r6: the library built without packed fp32 instructions: whole -m gpu suite, bench lines, training step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06hard}; mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/tests.log 2>&1; tail -5 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
sha256sum hamgnn_amd/lib/libhamgnn_hip.so | tee $out/lib.sha256
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench.err; python -c "
import json; r=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(round(r['value']), 'edges/s', round(r['ms_per_step'],2), 'ms frac', r['roofline']['frac'], 'complete', r.get('value_complete_programs'), 'acc', r.get('accuracy'))"
timeout 600 python tests/bench_training.py --workload si512 --steps 8 2>/dev/null | tail -2

#!/bin/bash
# round-end check of the committed state: whole -m gpu suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-final}; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py > $out/bench_sio2_10k_setA.json 2> $out/bench.err; tail -c 1500 $out/bench_sio2_10k_setA.json

#!/bin/bash
# whole -m gpu suite + smoke() (no bench line): the last check of a round when only the backward glue changed
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-finalt}; mkdir -p $out
timeout 200 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log

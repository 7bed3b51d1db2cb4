#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04m; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "linear_weight_gradient or backward or training or band_cal or export" > $out/tests.log 2>&1; tail -4 $out/tests.log
for f in 1 0; do HG_LINEAR_WGRAD=$f timeout 600 python tests/bench_training.py --workload si512 --steps 5 2>&1 | grep "^step 4" | sed "s/^/HG_LINEAR_WGRAD=$f /"; done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03am; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_properties or backward or training or twin" > $out/tests.log 2>&1; tail -8 $out/tests.log | cut -c1-300
python tests/bench_training.py --workload si512 --steps 5 2>&1 | tail -1

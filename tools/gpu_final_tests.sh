#!/bin/bash
# whole -m gpu suite exactly as the driver runs it (-x -q) + smoke() + which shared objects the test process loaded; one call = one lease (profiles/r06_final_validation.md)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-finalt}; mkdir -p $out
git_head=$(cat .git_head 2>/dev/null || echo "n/a")
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/tests.log 2>&1; tail -6 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
sha256sum hamgnn_amd/lib/libhamgnn_hip.so | tee $out/lib.sha256
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -2

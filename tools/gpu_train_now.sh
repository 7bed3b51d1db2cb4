# r6: where the training step stands (Si-512, set-A): step time, regions, the two radial-gradient products by library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06tr}; mkdir -p $out
timeout 600 python tests/bench_training.py --workload si512 --steps 8 > $out/train.log 2>&1; echo "train exit $?"; tail -3 $out/train.log
timeout 300 python tools/gpu_gs_gemm.py > $out/gemm.log 2>&1; cat $out/gemm.log
timeout 500 python tools/gpu_train_phases.py > $out/phases.log 2>&1; tail -22 $out/phases.log

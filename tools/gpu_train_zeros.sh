# round 5: the training step after the backward shortcuts (structural zeros in the adjoint / weight-gradient tables, compact radial GEMMs, one-launch head weight gradient)
cd /root/repo
out=gpurun_out/${1:-r05v}; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward or bit_reproducible or training or head or soc or refresh" > $out/tests.log 2>&1
echo "tests exit $?" >> $out/tests.log
tail -4 $out/tests.log
timeout 600 python tests/bench_training.py --workload si512 --steps 8 > $out/train.log 2>&1
echo "train exit $?"; tail -3 $out/train.log
timeout 500 python tools/gpu_train_phases.py > $out/phases.log 2>&1; tail -18 $out/phases.log

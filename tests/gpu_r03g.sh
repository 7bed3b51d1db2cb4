cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r03g; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lite_mode_and_zero_point or reference_loss_semantics or band_energy_loss_with_zero or bands_with_zero_point or two_rank_training or soc_su2_head_backward or head_soc_su2 or static_stream" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -40 $out/pytest.log

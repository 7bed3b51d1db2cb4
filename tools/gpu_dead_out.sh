cd /root/repo
mkdir -p gpurun_out/r05s
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unread_irreps or structural_zero or full_size or default_irreps or refresh" > gpurun_out/r05s/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r05s/tests.log
tail -5 gpurun_out/r05s/tests.log
timeout 600 python bench.py > gpurun_out/r05s/bench.json 2> gpurun_out/r05s/bench.err
echo "bench exit $?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r05s/bench.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["frac_full_program_launches"], r["roofline"]["frac_without_structural_zero_flops"])
print(r["roofline"]["launch_ms_by_position_in_step"], r["roofline"]["nonzero_flop_share_by_position"])
print({k: v for k, v in r.items() if "accuracy" in k})
PY

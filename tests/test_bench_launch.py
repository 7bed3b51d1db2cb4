"""bench.py as ONE command for N > 1: `python bench.py --gpus N` starts its own ranks under torch.distributed.run.  Without a GPU the ranks
cannot run -- what is checked here is the launcher leg: N ranks are started, each says who it is when it dies, the exit code is non-zero, no JSON line."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU twin is tests/test_gpu_parity.py::test_bench_script_two_rank_path_on_one_gpu[bare]")
def test_bare_command_launches_its_ranks_and_propagates_their_failure():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "si2", "--irreps", "B"],
                        capture_output=True, text=True, timeout=600, env=env)
    assert cp.returncode != 0
    fails = [l for l in cp.stderr.splitlines() if l.startswith("BENCH_RANK_FAILURE ")]      # (the launcher tears the other rank down as soon as the first one has died)
    assert 1 <= len(fails) <= 2 and all('"world_size": "2"' in f for f in fails), cp.stderr[-1500:]
    assert any(l.startswith("BENCH_LAUNCH_FAILURE ") for l in cp.stderr.splitlines())
    assert not [l for l in cp.stdout.splitlines() if l.startswith("{")]


def test_launcher_and_gpus_flag_must_agree():
    """under a launcher (WORLD_SIZE set) --gpus has to name the world size: a mismatch is refused before any device work"""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert cp.returncode != 0 and "WORLD_SIZE" in cp.stderr

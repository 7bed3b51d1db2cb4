import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # the fp64 oracle is made of many small torch ops: on the GPU box's 256 hardware threads the default intra-op pool makes them
    # ~100x slower than on 16 (one uni-chain test: 270 s instead of 3 s when no earlier test had capped the pool)
    import torch
    torch.set_num_threads(min(16, torch.get_num_threads()))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

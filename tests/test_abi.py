"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol include/hamgnn_hip.h declares."""
import os
import re

from hamgnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    _lib.build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "hamgnn_hip.h")).read()
    declared = set(re.findall(r"\b(hg_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.hg_version() >= 1


def test_product_does_not_import_oracle():
    import subprocess, sys
    code = "import sys; import hamgnn_amd, hamgnn_amd.models.hamgnn_conv, hamgnn_amd.models.hamgnn_output; assert not any(m.startswith('oracle') for m in sys.modules)"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hamgnn_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


def test_missing_gpu_fails_loudly():
    import pytest, torch
    from hamgnn_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ops._require_gpu(torch.zeros(1))

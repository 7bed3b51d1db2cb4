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


def test_scratch_query_is_host_only_and_matches_the_documented_sizes():
    """hg_scratch_bytes: the workspace query of the boundary (no entry point allocates); callable without a GPU"""
    import ctypes as C
    L = _lib.lib()
    L.hg_scratch_bytes.restype = C.c_int64
    L.hg_scratch_bytes.argtypes = [C.c_char_p, C.c_int64, C.c_int]
    assert L.hg_scratch_bytes(b"hg_edge_geometry", 822350, 0) == 822350 * 16          # ang_scratch [E][4] floats
    assert L.hg_scratch_bytes(b"hg_zero_point_shift", 1, 256) == 2 * 256 * 8            # partial_scratch [2 nparts] doubles
    assert all(L.hg_scratch_bytes(n.encode(), 1000, 0) == 0 for n in _lib.EXPORTS if n not in ("hg_edge_geometry", "hg_zero_point_shift"))
    assert L.hg_scratch_bytes(None, 1, 0) == -1 and L.hg_scratch_bytes(b"hg_tp_is", -1, 0) == -1

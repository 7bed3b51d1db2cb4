"""ctypes binding of libhamgnn_hip.so (the C ABI in include/hamgnn_hip.h).  There is NO CPU fallback: if the library is
missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HG_LIB_PATH", os.path.join(_HERE, "lib", "libhamgnn_hip.so"))   # override: kernel-variant A/B runs
_lib = None

EXPORTS = ["hg_last_error", "hg_version", "hg_scratch_bytes", "hg_edge_geometry", "hg_radial_basis", "hg_radial_hidden", "hg_rotate_gather", "hg_tp_fused", "hg_tp_is", "hg_tp_wgrad", "hg_row_program",
           "hg_segment_sum", "hg_gate", "hg_add_rows", "hg_to_planar", "hg_from_planar", "hg_embed_lookup", "hg_ham_merge",
           "hg_ham_finish", "hg_ham_readout", "hg_block_mean", "hg_soc_assemble", "hg_zero_point_shift", "hg_sym_contraction", "hg_sym_contraction3", "hg_hk_assemble",
           "hg_attn_logits", "hg_attn_aggregate", "hg_linear_planar", "hg_linear_wgrad", "hg_block_gemm", "hg_gate_backward", "hg_radial_hidden_multi", "hg_mfma_probe", "hg_build_config", "hg_norm_act", "hg_norm_act_backward", "hg_w3_split_refill"]


def build(verbose=False):
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libhamgnn_hip.so failed")
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the MI355X hot path has no CPU fallback)")
        import torch  # noqa: F401  -- first: the library must bind to the HIP runtime PyTorch already loaded (its bundled
        #                libamdhip64), not open a second copy of the runtime ("no ROCm-capable device" when loaded before torch)
        _lib = C.CDLL(LIB_PATH)
        _lib.hg_last_error.restype = C.c_char_p
        for name in EXPORTS:
            getattr(_lib, name)          # fail loudly if a declared symbol is not exported
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"{what}: libhamgnn_hip error {rc}: {lib().hg_last_error().decode()}")


def ptr(t):
    """device pointer of a (contiguous) torch tensor, or NULL."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "non-contiguous tensor handed to the C ABI"
    return C.c_void_p(t.data_ptr())


def i64(v):
    return C.c_int64(int(v))


def i32(v):
    return C.c_int(int(v))


def f32(v):
    return C.c_float(float(v))

"""Planar feature layout, item / segment record constants and the instruction table of the reference's tensor products (hamgnn_amd.plan: see the package docstring)."""
from __future__ import annotations

import os

import numpy as np

from ..so3 import Irreps

# item types
IT_TP = 0        # GEMM1 -> radial scale * CG coef -> GEMM2 -> add into segment tile
IT_LIN = 1       # GEMM1 only (plain o3.Linear path), rows = output channels, add into tile
IT_LINC = 2      # IT_LIN with a per-column coefficient (lite_mode uvu path: aligned-frame CG coefficient per m)
IT_POST = 3      # lite_mode segment post-op: tile <- Lc^T (s_e * tile)
IT_STREAM = 6    # lite_mode, input-stationary schedule (r4): the folded items of one PHASE as IS_WAVES_LITE balanced streams of uniform steps (plan._lite_streams)
LITE_SRING = int(os.environ.get("HG_LITE_SRING", "4"))   # request ring / descriptor block of csrc/tp_is.hip:stream_lite (SL_RING: 8 or 4)
IT_LINM = 4      # lite_mode, ALL paths (i, l_sh, k) of one (i, k) folded: one weight matrix per column, A_m = sum_paths cf_path[m] A_path (input-stationary kernel only)
# segment flags
SEG_UNROTATE = 1     # epilogue applies D^l(R_e)^T (messages go back to the global frame before the node scatter)

ITEM_I32 = 20        # int32 words per item record
SEG_I32 = 8          # int32 words per segment record
MAX_SRC = 4
STAGE_FLOATS = 2816      # = HG_STAGE_FLOATS of csrc/tp_fused.hip (wave-private LDS-DMA ring for B operands)


def ceil_div(a, b):
    return -(-a // b)


def rtm_max(nc):
    """row tiles per item by MM = (nc-1)/2: keeps the GEMM1 accumulators (rtm x nc f32x4 fragments) at <= 72 VGPRs, which is what
    the input-stationary kernel can hold next to its resident radial rows and double-buffered weight fragments without spilling
    (r1 table 4,4,4,3,2,2,1: 88 VGPRs, spilled; same MFMA count, 312 instead of 292 items for set-A)."""
    tab = [int(v) for v in os.environ.get("HG_RTM", "4,4,3,2,2,1,1").split(",")]
    return tab[(nc - 1) // 2]


class PlanarLayout:
    def __init__(self, irreps):
        self.irreps = Irreps(irreps)
        self.off, self.mulp = [], []
        o = 0
        for mul, l, p in self.irreps:
            mp = ceil_div(mul, 4) * 4
            self.off.append(o)
            self.mulp.append(mp)
            o += (2 * l + 1) * mp
        self.dim = o

    def index_map(self):
        """planar index of every e3nn-layout element: e3nn flat index -> planar flat index."""
        idx = np.zeros(self.irreps.dim, dtype=np.int64)
        e = 0
        for (mul, l, p), off, mp in zip(self.irreps, self.off, self.mulp):
            for u in range(mul):
                for a in range(2 * l + 1):
                    idx[e] = off + a * mp + u
                    e += 1
        return idx

    def to_planar(self, x):
        out = np.zeros(x.shape[:-1] + (self.dim,), dtype=x.dtype)
        out[..., self.index_map()] = x
        return out

    def from_planar(self, xp):
        return xp[..., self.index_map()]


def wigner_offsets(lmax):
    offs, o = [], 0
    for l in range(lmax + 1):
        offs.append(o)
        o += (2 * l + 1) ** 2
    return offs, o


# ------------------------------------------------------------------------------------------------ instruction tables


def tp_instructions(irreps1: Irreps, irreps2: Irreps, target: Irreps):
    """Reference rule (message_passing.py:147-171): one uvw path per (i, j, target entry with ir in ir_i x ir_j); output
    slots stably sorted by irrep; instructions re-ordered by sorted slot.  Returns list of (i, j, k_target, slot)."""
    slots, ins = [], []
    for i, (mi, li, pi) in enumerate(irreps1):
        for j, (_, lj, pj) in enumerate(irreps2):
            for k, (mk, lk, pk) in enumerate(target):
                if pk == pi * pj and abs(li - lj) <= lk <= li + lj:
                    ins.append((i, j, k, len(slots)))
                    slots.append((mk, lk, pk))
    _, perm = Irreps(slots).sort()
    ins = sorted([(i, j, k, perm[s]) for i, j, k, s in ins], key=lambda t: t[3])
    return ins


# ------------------------------------------------------------------------------------------------ small device tables

# e3nn normalize2mom constants (E_{z~N(0,1)}[act(z)^2]^(-1/2), e3nn's 1e6-sample Monte-Carlo recipe with CPU seed 0; values
# reproduced with torch 2.10).  Index = activation id of csrc/aux_kernels.hip:hg_act.
ACT_NONE, ACT_SSP, ACT_TANH, ACT_SILU, ACT_ABS = 0, 1, 2, 3, 4
ACT_CONSTS = np.array([1.0, 1.878204668541552, 1.5937334472592692, 1.6791767923989418, 1.0], dtype=np.float32)

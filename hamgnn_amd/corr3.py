"""Backward of the nu = 3 term of the MACE symmetric contraction (CorrProductBlock with `correlation: 3`; reference:
hamgnn/toolbox/mace/modules/symmetric_contraction.py:101-233, tools/cg.py:16-131).  Forward (hg_sym_contraction3, csrc/corr3.hip):

    out[n, o, c] += sum_{(x, i, j, kap, v) in ent3[o]} v W3[z_n, kap, c] h[n, x, c] h[n, i, c] h[n, j, c]

on top of the nu <= 2 part of hg_sym_contraction (plan.sym_contraction_tables lists the sparse entries of U_3 in the reference's path
order).  As for the nu <= 2 part (hamgnn_amd/backward_corr.py) the gradients of this node-level operation are gathers, products and
fixed-order segmented sums (ops.scatter_cols) on the device, in chunks of nodes sized to the entry list.  Device-agnostic: the CPU suite
checks them against autograd through the oracle's dense einsums."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import ops


def _entries3(tab: Dict, device):
    key = "_nu3_" + str(device)
    if key not in tab:
        g = lambda k: (tab[k].cpu().numpy() if torch.is_tensor(tab[k]) else np.asarray(tab[k]))
        ptr, ent = g("ptr3"), g("ent3")
        n = int(ptr[-1])
        o = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device).to(torch.int64)
        tab[key] = dict(o=t(o), x=t(ent[:n, 0]), i=t(ent[:n, 1]), j=t(ent[:n, 2]), k=t(ent[:n, 3]),
                        v=torch.as_tensor(np.ascontiguousarray(ent[:n, 4]).view(np.float32).copy(), device=device),
                        ell_off=t(g("ell_off")), out_off=t(g("out_off")))
    return tab[key]


def _chunk(E, C, chunk):
    """nodes per pass: the [n, entries, C] temporaries stay below ~2^25 elements each"""
    return max(1, min(int(chunk), (1 << 25) // max(1, int(E["o"].shape[0]) * C)))


def sym3_backward(tab: Dict, h: torch.Tensor, z: torch.Tensor, W3: torch.Tensor, C: int, g_out: torch.Tensor, chunk: int = 4096,
                  per_node: bool = False):
    """gradients of the nu = 3 term: (g_h [N, Dp], g_W3).  per_node: one weight block per node (z = arange(N), charge-doped attributes)."""
    E = _entries3(tab, h.device)
    ch = torch.arange(C, device=h.device)
    hcol = E["ell_off"][:, None] + ch[None, :]
    ocol = E["out_off"][:, None] + ch[None, :]
    g_h, gW3 = torch.zeros_like(h), torch.zeros_like(W3)
    zl = z.long()
    step = _chunk(E, C, chunk)
    v = E["v"][None, :, None].to(h.dtype)
    for n0 in range(0, h.shape[0], step):
        sl = slice(n0, min(h.shape[0], n0 + step))
        n = sl.stop - sl.start
        H = h[sl][:, hcol]
        G = g_out[sl][:, ocol]
        zc = zl[sl]
        nell = H.shape[1]
        t = G[:, E["o"]] * v                                      # [n, E3, C]
        hx, hi, hj = H[:, E["x"]], H[:, E["i"]], H[:, E["j"]]
        tw = t * W3[zc][:, E["k"]]
        gH = ops.scatter_cols(E["x"], tw * hi * hj, nell)
        gH = gH + ops.scatter_cols(E["i"], tw * hx * hj, nell)
        gH = gH + ops.scatter_cols(E["j"], tw * hx * hi, nell)
        p = ops.scatter_cols(E["k"], t * hx * hi * hj, W3.shape[1])
        if per_node:
            gW3[sl] = p
        else:
            gW3 += ops.scatter_rows(zc, p, W3.shape[0], persistent=False)
        g_h[sl] = g_h[sl].index_add(1, hcol.reshape(-1), gH.reshape(n, -1))
    return g_h, gW3

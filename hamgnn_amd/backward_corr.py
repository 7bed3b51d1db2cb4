"""Backward of the MACE symmetric contraction of a CorrProductBlock (correlation 2; SURVEY 8f-3 x a21; reference:
hamgnn/toolbox/mace/modules/symmetric_contraction.py:101-233, hamgnn/nn/interaction_blocks.py:234-260).

Forward (hg_sym_contraction, plan.sym_contraction_tables), per node n, channel c and output element o = (target irrep, component):
    out[n, o, c] = sum_{(x, kap, v) in ent1[o]} v W1[z_n, kap, c] h[n, x, c]
                 + sum_{(x, i, kap, v) in ent2[o]} v W2[z_n, kap, c] h[n, i, c] h[n, x, c]
A node-level operation (N rows of a few hundred floats): the gradients with respect to h, W1 and W2 are gathers, products and
fixed-order segmented sums (ops.scatter_cols) over the sparse entry lists -- torch tensor ops on the device, in chunks of nodes (first version; the forward kernel's
loop nest with the roles of `out` and `h` exchanged is the HIP form).  Device-agnostic, so the CPU suite checks it against autograd
through the oracle's dense einsums."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import ops


def _entries(tab: Dict, device):
    """flat (output element, ...) index tensors of the two sparse tables, cached on the table dict per device"""
    key = "_bw_" + str(device)
    if key not in tab:
        g = lambda k: (tab[k].cpu().numpy() if torch.is_tensor(tab[k]) else np.asarray(tab[k]))
        ptr1, ent1, ptr2, ent2 = g("ptr1"), g("ent1"), g("ptr2"), g("ent2")
        n1, n2 = int(ptr1[-1]), int(ptr2[-1])
        o1 = np.repeat(np.arange(len(ptr1) - 1), np.diff(ptr1))
        o2 = np.repeat(np.arange(len(ptr2) - 1), np.diff(ptr2))
        t = lambda a, dt=torch.int64: torch.as_tensor(np.ascontiguousarray(a), device=device).to(dt)
        val = lambda e, n: torch.as_tensor(np.ascontiguousarray(e[:n, 3]).view(np.float32).copy(), device=device)
        tab[key] = dict(o1=t(o1), x1=t(ent1[:n1, 0]), k1=t(ent1[:n1, 1]), v1=val(ent1, n1),
                        o2=t(o2), x2=t(ent2[:n2, 0]), i2=t(ent2[:n2, 1]), k2=t(ent2[:n2, 2]), v2=val(ent2, n2),
                        ell_off=t(g("ell_off")), out_off=t(g("out_off")))
    return tab[key]


def sym_contraction_backward(tab: Dict, h: torch.Tensor, z: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor, C: int, g_out: torch.Tensor,
                             chunk: int = 4096, per_node: bool = False):
    """h, g_out: planar hidden rows [N, Dp]; W1 [nel, K1, C], W2 [nel, K2, C].  Returns (g_h [N, Dp], g_W1, g_W2).
    per_node: W1 / W2 hold one weight block PER NODE (z = arange(N): the charge-doped attributes mix the element blocks, nn.CorrProductBlock)."""
    E = _entries(tab, h.device)
    N, Dp = h.shape
    dt = h.dtype
    ch = torch.arange(C, device=h.device)
    hcol = (E["ell_off"][:, None] + ch[None, :])                  # [num_ell, C] planar columns of h
    ocol = (E["out_off"][:, None] + ch[None, :])                  # [nout, C]
    g_h = torch.zeros_like(h)
    gW1, gW2 = torch.zeros_like(W1), torch.zeros_like(W2)
    zl = z.long()
    for n0 in range(0, N, chunk):
        sl = slice(n0, min(N, n0 + chunk))
        n = sl.stop - sl.start
        H = h[sl][:, hcol]                                        # [n, num_ell, C]
        G = g_out[sl][:, ocol]                                    # [n, nout, C]
        zc = zl[sl]
        nell = H.shape[1]
        # every reduction in a fixed order (ops.scatter_cols / scatter_rows: sort + segmented sum) -- index_add_'s float atomics would make
        # a training step with the CorrProductBlock differ from run to run
        # nu = 1
        t1 = G[:, E["o1"]] * E["v1"][None, :, None].to(dt)        # [n, E1, C]
        gH = ops.scatter_cols(E["x1"], t1 * W1[zc][:, E["k1"]], nell)
        p1 = ops.scatter_cols(E["k1"], t1 * H[:, E["x1"]], W1.shape[1])
        if per_node:
            gW1[sl] = p1
        else:
            gW1 += ops.scatter_rows(zc, p1, W1.shape[0], persistent=False)
        # nu = 2
        t2 = G[:, E["o2"]] * E["v2"][None, :, None].to(dt)        # [n, E2, C]
        hx, hi = H[:, E["x2"]], H[:, E["i2"]]
        tw = t2 * W2[zc][:, E["k2"]]
        gH = gH + ops.scatter_cols(E["x2"], tw * hi, nell)
        gH = gH + ops.scatter_cols(E["i2"], tw * hx, nell)
        p2 = ops.scatter_cols(E["k2"], t2 * hx * hi, W2.shape[1])
        if per_node:
            gW2[sl] = p2
        else:
            gW2 += ops.scatter_rows(zc, p2, W2.shape[0], persistent=False)
        g_h[sl].index_add_(1, hcol.reshape(-1), gH.reshape(n, -1))     # (distinct columns: nothing is summed here)
    return g_h, gW1, gW2

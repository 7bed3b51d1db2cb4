"""Backward of the attention aggregation of an AttentionBlockE3 (SURVEY 8f-3 x 8f-4; reference: hamgnn/nn/attention.py:91-164, 339-352;
forward kernels: csrc/attention.hip hg_attn_logits / hg_attn_aggregate):

    logit[e, h] = cut(r_e) / sqrt(head_dim) * sum_{col in head h} K[src_e, col] K[dst_e, col]      cut(r) = soft_unit_step(p (1 - r / r_c))
    alpha[e, h] = exp(logit - max_dst) / (sum_dst exp(...) + 1e-16)                                 (PyG softmax over the incoming edges)
    agg[n, col] = sum_{e: dst_e = n} alpha[e, head(col)] V[e, col]

Gradients with respect to the key rows K [N, Dp], the value rows V [E, Dp] and the learnable cutoff parameter p, as gathers, two small
GEMMs against the column -> head indicator and fixed-order row scatters (`ops.scatter_rows`: stable sort + segmented sum, no float atomics; torch tensor ops on the device; edge-level, HBM-bound: the HIP form is the
two forward kernels with the roles of `agg` and `V` exchanged -- first version).  Device-agnostic: the CPU suite checks it against
autograd through the oracle."""
from __future__ import annotations

import math

import torch

from . import ops


def attention_backward(K: torch.Tensor, V: torch.Tensor, g_agg: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, length: torch.Tensor,
                       head_tab: torch.Tensor, num_heads: int, head_dim: int, cut_param: torch.Tensor, cutoff: float, allreduce=None):
    """K, g_agg: [N, Dp] planar node rows; V: [E, Dp] planar value rows (the frame the forward aggregated them in); head_tab: int [Dp]
    (head of a column, -1 = padding).  Returns (g_K [N, Dp], g_V [E, Dp], g_cut_param [scalar tensor]).
    allreduce: for an EDGE-SHARDED graph, `allreduce(tensor, op)` with op in {"max", "sum"} (in place, over the ranks): the soft-max
    statistics, the per-node dot products, the key gradient and the cutoff gradient are then those of all edges (V, g_V stay local)."""
    N, Dp = K.shape
    dt = K.dtype
    src, dst = src.long(), dst.long()
    M = torch.zeros(Dp, num_heads, device=K.device, dtype=dt)                  # column -> head indicator (padding columns: all zero)
    cols = torch.nonzero(head_tab >= 0).reshape(-1)
    M[cols, head_tab[cols].long()] = 1.0
    Ks, Kd, Gd = K[src], K[dst], g_agg[dst]
    scale = 1.0 / math.sqrt(head_dim)
    p = cut_param.reshape(()).to(dt)
    u = 1.0 - length.to(dt) / cutoff
    x = p * u
    pos = x > 0
    xs = torch.where(pos, x, torch.ones_like(x))
    cut = torch.where(pos, torch.exp(-1.0 / xs), torch.zeros_like(x))          # [E]
    Dh = (Ks * Kd) @ M                                                          # [E, H]
    logit = cut[:, None] * scale * Dh
    mx = torch.full((N, num_heads), -float("inf"), device=K.device, dtype=dt).scatter_reduce(0, dst[:, None].expand(-1, num_heads), logit, "amax")
    if allreduce is not None:
        allreduce(mx, "max")
    ex = torch.exp(logit - mx[dst])
    zsum = ops.scatter_rows(dst, ex, N)
    if allreduce is not None:
        allreduce(zsum, "sum")
    alpha = ex / (zsum + 1e-16)[dst]                                            # [E, H]
    g_V = (alpha @ M.t()) * Gd
    g_alpha = (V * Gd) @ M                                                      # [E, H]
    dot = ops.scatter_rows(dst, alpha * g_alpha, N)
    if allreduce is not None:
        allreduce(dot, "sum")
    g_logit = alpha * (g_alpha - dot[dst])
    g_D = (g_logit * cut[:, None] * scale) @ M.t()                              # [E, Dp], per column of its head
    g_K = ops.scatter_rows(src, g_D * Kd, N) + ops.scatter_rows(dst, g_D * Ks, N)      # fixed summation order (no float atomics)
    g_cut = (g_logit * Dh).sum(1) * scale                                       # [E]
    g_p = (g_cut * torch.where(pos, cut / (xs * xs), torch.zeros_like(x)) * u).sum().reshape(1)
    if allreduce is not None:
        allreduce(g_K, "sum")
        allreduce(g_p, "sum")
    return g_K, g_V, g_p.reshape(())

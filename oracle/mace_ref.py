"""oracle/mace_ref.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Never imported by the product path.

Restatement of the (optional, off in the shipped configs) correlation-product node update of the reference:
  CorrProductBlock                 hamgnn/nn/interaction_blocks.py:168-260
  EquivariantProductBasisBlock     hamgnn/toolbox/mace/modules/blocks.py:171-207
  SymmetricContraction/Contraction hamgnn/toolbox/mace/modules/symmetric_contraction.py:29-233
  U_matrix_real / _wigner_nj       hamgnn/toolbox/mace/tools/cg.py:16-131
  reshape_irreps                   hamgnn/toolbox/mace/modules/irreps_tools.py:67-87
on top of oracle/e3.py.  Parameter names equal the reference's, so state_dicts are interchangeable (oracle/gen_golden.py
checks that against the reference's own modules, in the build container).  PARITY STATUS: as oracle/hamgnn_ref.py.
"""
from __future__ import annotations

from typing import List

import torch
from torch import nn

from . import e3
from .e3 import Irrep, Irreps, Linear


def wigner_nj(irrepss: List[Irreps], dtype=None):
    """cg.py:16-87 with normalization='component', no mid filter: list of (ir_out, C) sorted by ir_out, C of shape
    [ir_out.dim, irreps_1.dim, ..., irreps_n.dim]."""
    irrepss = [Irreps(x) for x in irrepss]
    if len(irrepss) == 1:
        (irreps,) = irrepss
        ret, eye, i = [], torch.eye(irreps.dim, dtype=dtype), 0
        for mul, ir in irreps:
            for _ in range(mul):
                ret.append((ir, eye[i:i + ir.dim]))
                i += ir.dim
        return ret
    *left, right = irrepss
    ret = []
    for ir_left, C_left in wigner_nj(left, dtype=dtype):
        i = 0
        for mul, ir in right:
            for ir_out in ir_left * ir:
                C = e3.wigner_3j(ir_out.l, ir_left.l, ir.l, dtype=dtype) * ir_out.dim ** 0.5
                C = torch.einsum("jk,ijl->ikl", C_left.flatten(1), C)
                C = C.reshape(ir_out.dim, *(x.dim for x in left), ir.dim)
                for u in range(mul):
                    E = torch.zeros(ir_out.dim, *(x.dim for x in left), right.dim, dtype=dtype)
                    E[..., i + u * ir.dim:i + (u + 1) * ir.dim] = C
                    ret.append((ir_out, E))
            i += mul * ir.dim
    return sorted(ret, key=lambda t: t[0])                     # stable, by irrep only (cg.py:87)


def u_matrix_real(irreps_in, ir_out: Irrep, correlation: int, dtype=None) -> torch.Tensor:
    """cg.py:90-131 for a single target irrep: all couplings irreps_in^(x nu) -> ir_out stacked on the last axis (squeezed)."""
    stack = None
    for ir, C in wigner_nj([Irreps(irreps_in)] * correlation, dtype=dtype):
        if ir == ir_out:
            c = C.squeeze().unsqueeze(-1)
            stack = c if stack is None else torch.cat((stack, c), dim=-1)
    if stack is None:
        raise ValueError(f"{ir_out} is not reachable from {irreps_in} at correlation {correlation}")
    return stack


class Contraction(nn.Module):
    """symmetric_contraction.py:101-233 for any correlation: the reference generates one einsum per nu (main: "[w] x.. i k, ekc, bci, be ->
    bc [w] x.."; then per lower nu "[w] x.. k, ekc, be -> bc [w] x.." added to the running tensor, whose last component index is
    contracted with x again) and lets opt_einsum_fx reorder them; here they are evaluated as written."""

    def __init__(self, irreps_in: Irreps, ir_out: Irrep, correlation: int, num_elements: int):
        super().__init__()
        assert 1 <= correlation <= 4
        self.num_features = sum(mul for mul, ir in irreps_in if ir.l == 0 and ir.p == 1)
        coupling = Irreps([(1, ir) for _, ir in irreps_in])
        self.correlation, self.scalar_out = correlation, ir_out.l == 0
        dtype = torch.get_default_dtype()
        for nu in range(1, correlation + 1):
            self.register_buffer(f"U_matrix_{nu}", u_matrix_real(coupling, ir_out, nu, dtype=dtype))
        self.weights_max = nn.Parameter(torch.randn(num_elements, self.U(correlation).shape[-1], self.num_features) / self.U(correlation).shape[-1])
        self.weights = nn.ParameterList([nn.Parameter(torch.randn(num_elements, self.U(nu).shape[-1], self.num_features) / self.U(nu).shape[-1])
                                         for nu in range(correlation - 1, 0, -1)])

    def U(self, nu):
        return getattr(self, f"U_matrix_{nu}")

    def forward(self, x, y):
        """x: [b, c, num_ell]; y: node attributes [b, num_elements] (one-hot, or the charge-doped mixture) -> [b, c * (2L+1)]"""
        w = "" if self.scalar_out else "w"
        free = "xvuts"[:self.correlation - 1]                       # the component indices that stay open after the main contraction
        out = torch.einsum(f"{w}{free}ik,ekc,bci,be->bc{w}{free}", self.U(self.correlation), self.weights_max, x, y)
        for j, weight in enumerate(self.weights):                   # nu = correlation - 1 ... 1
            nu = self.correlation - 1 - j
            idx = "xvuts"[:nu]
            c = torch.einsum(f"{w}{idx}k,ekc,be->bc{w}{idx}", self.U(nu), weight, y) + out
            out = torch.einsum(f"bc{w}{idx[:-1]}i,bci->bc{w}{idx[:-1]}", c, x)
        return out.reshape(out.shape[0], -1)


class SymmetricContraction(nn.Module):
    def __init__(self, irreps_in, irreps_out, correlation, num_elements):
        super().__init__()
        irreps_in, irreps_out = Irreps(irreps_in), Irreps(irreps_out)
        self.contractions = nn.ModuleList([Contraction(irreps_in, ir, correlation, num_elements) for _, ir in irreps_out])

    def forward(self, x, y):
        return torch.cat([c(x, y) for c in self.contractions], dim=-1)


class EquivariantProductBasisBlock(nn.Module):
    def __init__(self, node_feats_irreps, target_irreps, correlation, num_elements):
        super().__init__()
        self.symmetric_contractions = SymmetricContraction(node_feats_irreps, target_irreps, correlation, num_elements)
        self.linear = Linear(target_irreps, target_irreps)

    def forward(self, node_feats, node_attrs):
        return self.linear(self.symmetric_contractions(node_feats, node_attrs))


def reshape_irreps(irreps: Irreps, x):
    """[b, sum mul*d] -> [b, mul, sum d] (irreps_tools.py:67-87; all multiplicities equal)."""
    out, ix = [], 0
    for mul, ir in Irreps(irreps):
        out.append(x[:, ix:ix + mul * ir.dim].reshape(x.shape[0], mul, ir.dim))
        ix += mul * ir.dim
    return torch.cat(out, dim=-1)


class CorrProductBlock(nn.Module):
    def __init__(self, irreps_node_feats, num_hidden_features, correlation, num_elements, use_skip_connections=True):
        super().__init__()
        self.irreps_node_feats = Irreps(irreps_node_feats).simplify()
        self.irreps_hidden = Irreps([(num_hidden_features, ir) for _, ir in self.irreps_node_feats])
        self.use_skip_connections = use_skip_connections
        self.linear_pre = Linear(self.irreps_node_feats, self.irreps_hidden)
        self.linear_sc = Linear(self.irreps_node_feats, self.irreps_node_feats)
        self.prod = EquivariantProductBasisBlock(self.irreps_hidden, self.irreps_hidden, correlation, num_elements)
        self.linear_out = Linear(self.irreps_hidden, self.irreps_node_feats)

    def forward(self, node_features, node_attrs):
        """returns the NEW node features (the reference stores them into the graph dict, interaction_blocks.py:255-258)"""
        h = reshape_irreps(self.irreps_hidden, self.linear_pre(node_features))
        out = self.linear_out(self.prod(h, node_attrs))
        return out + self.linear_sc(node_features) if self.use_skip_connections else out

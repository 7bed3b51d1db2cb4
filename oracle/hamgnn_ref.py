"""oracle/hamgnn_ref.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Never imported by the product path.

Unfused pure-torch restatement (one einsum per e3nn instruction, materialised ``mid`` tensors, index_add scatter)
of the HamGNN hot path on top of oracle/e3.py.  Parameter/attribute names equal the reference's so state_dicts
are interchangeable with the reference modules (oracle/gen_golden.py checks exactly that, in this container).

Each class cites the reference code it follows (paths relative to /root/reference):
  tp_instructions                  hamgnn/nn/message_passing.py:136-171  (== tensor_products.py:115-150)
  LinearScaleWithWeights           hamgnn/nn/tensor_products.py:25-47
  RadialTensorProduct              hamgnn/nn/tensor_products.py:51-189   (TensorProductWithMemoryOptimizationWithWeight)
  MessagePackBlock                 hamgnn/nn/message_passing.py:26-231
  ResidualBlock                    hamgnn/nn/interaction_blocks.py:264-358 ; irreps2gate hamgnn/utils/irreps_utils.py:33-65
  ConvBlockE3                      hamgnn/nn/convolution.py:22-160
  PairInteractionBlock             hamgnn/nn/interaction_blocks.py:30-164
  PairInteractionEmbeddingBlock    hamgnn/nn/embeddings.py:215-337
  edge geometry                    hamgnn/toolbox/nequip/nn/embedding/_edge.py:59-67 ; hamgnn/nn/embeddings.py:73-100 ;
                                   hamgnn/utils/basis_functions.py:177-208 ; hamgnn/utils/cutoff_functions.py:35-61
  HamGNNConvE3                     hamgnn/models/hamgnn_conv.py:88-284
  AttentionAggregation / AttentionBlockE3 / HamGNNTransformer
                                   hamgnn/nn/attention.py:91-360 ; hamgnn/nn/attention_utils.py ; hamgnn/models/hamgnn_transformer.py:36-250
  HamLayer / HamGNNPlusPlusOut     hamgnn/models/hamgnn_output.py:38-58, 96-343, 851-891, 1056-1096, 1187-1285,
                                   2288-2365, 2916-3000, 3026-3144 (SOC so3), 3772-3799, 3966-4003

PARITY STATUS: "parity unpinned" against e3nn (see oracle/e3.py header); HamGNN-level wiring pinned by
tests/golden/*.npz generated from the reference's own modules (oracle/gen_golden.py).
"""
from __future__ import annotations

import json
import numpy as np
import math
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from . import e3
from .e3 import Irreps, Irrep, Linear, TensorProduct, FullyConnectedNet, Gate

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def ssp(x):
    return torch.nn.functional.softplus(x) - math.log(2.0)


ssp.__name__ = "ShiftedSoftPlus"
ACTS = {"abs": torch.abs, "tanh": torch.tanh, "ssp": ssp, "silu": torch.nn.functional.silu}


def scatter_sum(src, index, dim_size):
    out = src.new_zeros((dim_size,) + src.shape[1:])
    return out.index_add_(0, index, src)


# ---------------------------------------------------------------------------------------------- tensor products


def tp_instructions(irreps1, irreps2, target, mode="uvw", trainable=True):
    irreps1, irreps2, target = Irreps(irreps1), Irreps(irreps2), Irreps(target)
    slots, ins = [], []
    for i, (mul_in, ir_in) in enumerate(irreps1):
        for j, (_, ir_sh) in enumerate(irreps2):
            prod = ir_in * ir_sh
            for mul_out, ir_out in target:
                if ir_out in prod:
                    ins.append((i, j, len(slots), mode, trainable))
                    slots.append((mul_out if mode == "uvw" else mul_in, ir_out))
    mid, perm, _ = Irreps(slots).sort()
    ins = sorted([(a, b, perm[c], m, t) for a, b, c, m, t in ins], key=lambda x: x[2])
    return mid, ins


class LinearScaleWithWeights(nn.Module):
    def __init__(self, irreps_in, irreps_out):
        super().__init__()
        irreps_in = Irreps(irreps_in)
        self.tp = TensorProduct(irreps_in, "1x0e", irreps_in, [(i, 0, i, "uvu", True) for i in range(len(irreps_in))],
                                shared_weights=False, internal_weights=False)
        self.weight_numel = self.tp.weight_numel
        self.linear_out = Linear(irreps_in, irreps_out)

    def forward(self, x, weight):
        return self.linear_out(self.tp(x, torch.ones_like(x[:, :1]), weight))


class RadialTensorProduct(nn.Module):
    def __init__(self, irreps_input_1, irreps_input_2, irreps_out, irreps_scalar, radial_MLP, lite_mode=False):
        super().__init__()
        mode = "uvu" if lite_mode else "uvw"
        self.irreps_mid, self.instructions = tp_instructions(irreps_input_1, irreps_input_2, irreps_out, mode, not lite_mode)
        self.tensor_product = TensorProduct(irreps_input_1, irreps_input_2, self.irreps_mid, self.instructions,
                                            internal_weights=True, shared_weights=True)
        self.linear_scaler = LinearScaleWithWeights(self.irreps_mid.simplify(), irreps_out)
        self.weight_generator = FullyConnectedNet([Irreps(irreps_scalar).num_irreps] + list(radial_MLP) + [self.linear_scaler.weight_numel],
                                                  torch.nn.functional.silu)

    def forward(self, x, y, scalars):
        return self.linear_scaler(self.tensor_product(x, y), self.weight_generator(scalars))


def heads_to_vector(irreps, x):
    """[E, 2, D] -> per irrep block [src block, dst block]  (hamgnn/nn/attention_utils.py:85-119)."""
    parts, i = [], 0
    for mul, ir in Irreps(irreps):
        n = mul * ir.dim
        parts.append(x[:, :, i:i + n].reshape(x.shape[0], -1))
        i += n
    return torch.cat(parts, dim=1)


class MessagePackBlock(nn.Module):
    def __init__(self, irreps_node_feats, irreps_edge_feats, irreps_local_env_edge, irreps_out, irreps_edge_scalars,
                 radial_MLP=(64, 64), lite_mode=False):
        super().__init__()
        self.irreps_node_feats = Irreps(irreps_node_feats)
        irreps_edge_feats, irreps_sh, irreps_out = Irreps(irreps_edge_feats), Irreps(irreps_local_env_edge), Irreps(irreps_out)
        self.lite_mode = lite_mode
        mode = "uvu" if lite_mode else "uvw"
        comb = Irreps([(max(1, int(mul * 2)), ir) for mul, ir in self.irreps_node_feats])
        self.mid_node_irreps, self.node_instructions = tp_instructions(comb, irreps_sh, irreps_out, mode, not lite_mode)
        self.mid_edge_irreps, self.edge_instructions = tp_instructions(irreps_edge_feats, irreps_sh, irreps_out, mode, not lite_mode)
        self.node_tensor_product = TensorProduct(comb, irreps_sh, self.mid_node_irreps, self.node_instructions,
                                                 internal_weights=True, shared_weights=True)
        self.edge_tensor_product = TensorProduct(irreps_edge_feats, irreps_sh, self.mid_edge_irreps, self.edge_instructions,
                                                 internal_weights=True, shared_weights=True)
        n_in = Irreps(irreps_edge_scalars).num_irreps
        if lite_mode:
            self.node_linear_scaler = Linear(self.mid_node_irreps.simplify(), irreps_out)
            self.edge_linear_scaler = Linear(self.mid_edge_irreps.simplify(), irreps_out)
            self.combine_messages = LinearScaleWithWeights(irreps_out.simplify(), irreps_out)
            self.weight_generator_combine = FullyConnectedNet([n_in] + list(radial_MLP) + [self.combine_messages.weight_numel],
                                                              torch.nn.functional.silu)
        else:
            self.node_linear_scaler = LinearScaleWithWeights(self.mid_node_irreps.simplify(), irreps_out)
            self.edge_linear_scaler = LinearScaleWithWeights(self.mid_edge_irreps.simplify(), irreps_out)
            self.node_weight_generator = FullyConnectedNet([n_in] + list(radial_MLP) + [self.node_linear_scaler.weight_numel],
                                                           torch.nn.functional.silu)
            self.edge_weight_generator = FullyConnectedNet([n_in] + list(radial_MLP) + [self.edge_linear_scaler.weight_numel],
                                                           torch.nn.functional.silu)
            self.node_linear_out = Linear(irreps_out, irreps_out)
            self.edge_linear_out = Linear(irreps_out, irreps_out)

    def forward(self, src, dst, edge_feats, sh, rbf):
        node_inter = heads_to_vector(self.irreps_node_feats, torch.stack([src, dst], dim=-2))
        if self.lite_mode:
            a = self.node_linear_scaler(self.node_tensor_product(node_inter, sh))
            b = self.edge_linear_scaler(self.edge_tensor_product(edge_feats, sh))
            return self.combine_messages(a + b, self.weight_generator_combine(rbf))
        a = self.node_linear_scaler(self.node_tensor_product(node_inter, sh), self.node_weight_generator(rbf))
        b = self.edge_linear_scaler(self.edge_tensor_product(edge_feats, sh), self.edge_weight_generator(rbf))
        return self.node_linear_out(a) + self.edge_linear_out(b)


# ---------------------------------------------------------------------------------------------- gate / residual


def irreps2gate(irreps, act_scalars={1: "ssp", -1: "tanh"}, act_gates={1: "ssp", -1: "abs"}):
    irreps = Irreps(irreps)
    scalars = Irreps([(m, ir) for m, ir in irreps if ir.l == 0]).simplify()
    gated = Irreps([(m, ir) for m, ir in irreps if ir.l != 0]).simplify()
    gates = Irreps([(m, "0e") for m, _ in gated]).simplify() if gated.dim > 0 else Irreps([])
    return scalars, gates, gated, [ACTS[act_scalars[ir.p]] for _, ir in scalars], [ACTS[act_gates[ir.p]] for _, ir in gates]


class ResidualBlock(nn.Module):
    def __init__(self, irreps_in, feature_irreps_hidden, resnet=True, nonlinearity_type="gate"):
        """interaction_blocks.py:262-358; nonlinearity_type "norm": e3nn NormActivation with the even-scalar nonlinearity (ssp) AS GIVEN, normalize=True,
        epsilon=1e-8, no bias (:311-330)"""
        super().__init__()
        assert nonlinearity_type in ("gate", "norm")
        if nonlinearity_type == "norm":
            from .e3 import NormActivation
            self.equivariant_nonlin = NormActivation(Irreps(feature_irreps_hidden), ACTS["ssp"], normalize=True, epsilon=1e-8, bias=False)
        else:
            s, g, gd, a_s, a_g = irreps2gate(feature_irreps_hidden)
            self.equivariant_nonlin = Gate(s, a_s, g, a_g, gd)
        self.linear1 = Linear(irreps_in, self.equivariant_nonlin.irreps_in)
        self.linear2 = Linear(self.equivariant_nonlin.irreps_out, irreps_in)
        self.resnet = resnet

    def forward(self, x):
        y = self.linear2(self.equivariant_nonlin(self.linear1(x)))
        return x + y if self.resnet else y


# ---------------------------------------------------------------------------------------------- backbone blocks


class ConvBlockE3(nn.Module):
    def __init__(self, irreps_in, irreps_out, irreps_edge_attrs, irreps_edge_embed, radial_MLP, lite_mode=False):
        super().__init__()
        self.residual = ResidualBlock(irreps_in, irreps_out)
        self.conv_tp = MessagePackBlock(irreps_in, irreps_in, irreps_edge_attrs, irreps_out, irreps_edge_embed, radial_MLP, lite_mode)
        self.skip_linear = Linear(irreps_in, irreps_out)

    def forward(self, g):
        sender, receiver = g["edge_index"]
        x = g["node_features"]
        skip = self.skip_linear(x)
        msg = self.conv_tp(x[sender], x[receiver], g["edge_features"], g["edge_attrs"], g["edge_embedding"])
        agg = scatter_sum(msg, receiver, x.shape[0])
        g["node_features"] = self.residual(agg) + skip
        return g["node_features"]


class PairInteractionBlock(nn.Module):
    def __init__(self, irreps_node_feats, irreps_edge_attrs, irreps_edge_embed, irreps_edge_feats, radial_MLP,
                 use_skip_connections=True, legacy_edge_update=False, lite_mode=False):
        super().__init__()
        self.use_skip_connections, self.legacy_edge_update = use_skip_connections, legacy_edge_update
        self.linear_up_src = Linear(irreps_node_feats, irreps_node_feats)
        self.linear_up_tar = Linear(irreps_node_feats, irreps_node_feats)
        self.conv_tp = MessagePackBlock(irreps_node_feats, irreps_edge_feats, irreps_edge_attrs, irreps_edge_feats,
                                        irreps_edge_embed, radial_MLP, lite_mode)
        if use_skip_connections:
            self.skip_linear = Linear(irreps_edge_feats, irreps_edge_feats)

    def forward(self, g):
        src, dst = g["edge_index"]
        x, f = g["node_features"], g["edge_features"]
        mix = self.conv_tp(self.linear_up_src(x)[src], self.linear_up_tar(x)[dst], f, g["edge_attrs"], g["edge_embedding"])
        if self.use_skip_connections:
            f = mix + self.skip_linear(f)
        elif not self.legacy_edge_update:
            f = mix
        g["edge_features"] = f
        return f


class PairInteractionEmbeddingBlock(nn.Module):
    def __init__(self, irreps_node_attrs, irreps_edge_attrs, irreps_edge_embed, irreps_edge_feats, radial_MLP, lite_mode=False):
        super().__init__()
        self.linear_up_src = Linear(irreps_node_attrs, irreps_node_attrs)
        self.linear_up_dst = Linear(irreps_node_attrs, irreps_node_attrs)
        self.conv_tp = RadialTensorProduct(irreps_node_attrs, irreps_edge_attrs, irreps_edge_feats, irreps_edge_embed, radial_MLP, lite_mode)

    def forward(self, g):
        src, dst = g["edge_index"]
        a = g["node_features"]
        x = self.linear_up_src(a[src]) + self.linear_up_dst(a[dst])
        g["edge_features"] = self.conv_tp(x, g["edge_attrs"], g["edge_embedding"])
        return g["edge_features"]


class _Holder(nn.Module):
    pass


def edge_geometry(pos, edge_index, nbr_shift, irreps_sh, cutoff, num_radial, sh_normalize=True, sh_normalization="component", rbf_func="bessel"):
    """edge_attrs (SH of v[[1,2,0]]), edge_embedding (radial basis * cosine cutoff: hamgnn/nn/embeddings.py:73-100), lengths.
    j = row 0, i = row 1.  rbf_func: "bessel" (utils/basis_functions.py:177-208) or "gaussian" (GaussianSmearing, :211-224, start 0,
    stop = cutoff; the width comes from the fp32 linspace, as the reference's `.item()` of a float32 tensor)."""
    j, i = edge_index
    vec = (pos[i] + nbr_shift) - pos[j]
    unit = torch.nn.functional.normalize(vec, dim=-1)
    sh = e3.spherical_harmonics(Irreps(irreps_sh).ls, unit[:, [1, 2, 0]], sh_normalize, sh_normalization)
    r = vec.norm(dim=-1)
    if rbf_func == "bessel":
        freqs = torch.arange(1, num_radial + 1, dtype=pos.dtype) * math.pi / cutoff
        rbf = torch.sin(r[:, None] * freqs[None, :]) / r[:, None]
    elif rbf_func == "gaussian":
        offset32 = torch.linspace(0.0, cutoff, num_radial, dtype=torch.float32)
        coeff = -0.5 / (offset32[1] - offset32[0]).item() ** 2
        rbf = torch.exp(coeff * (r[:, None] - offset32.to(pos.dtype)[None, :]) ** 2)
    else:
        raise NotImplementedError(rbf_func)
    fc = 0.5 * (torch.cos(r * math.pi / cutoff) + 1.0) * (r < cutoff).to(pos.dtype)
    return sh, rbf * fc[:, None], r


class Embedding_block_q(nn.Module):
    """charge-doping node attributes (hamgnn/toolbox/nequip/nn/embedding/_embedding_block.py:56-137; mlp_q = denseRegression(n_h=2,
    Softplus, no batch norm), hamgnn/utils/regression_layers.py:7-21, utils/mlp.py:11-35): one_hot(z) + mlp_q(gauss(q)) - mlp_q(gauss(0))"""

    def __init__(self, num_types, num_charge_attr_feas):
        super().__init__()
        F = num_charge_attr_feas
        self.charge_min, self.charge_max = -8.0, 8.0
        width = (self.charge_max - self.charge_min) / (F - 1) if F > 1 else 1.0
        centers = torch.linspace(self.charge_min, self.charge_max, steps=F)
        self.register_buffer("charge_centers", centers)
        self.register_buffer("charge_gamma", torch.tensor(1.0 / width ** 2))
        self.register_buffer("neutral_charge_attrs", torch.exp(-(1.0 / width ** 2) * centers * centers).view(1, -1))
        self.mlp_q = _Holder()
        self.mlp_q.fcs = nn.ModuleList([nn.Sequential(nn.Linear(F, F, bias=True), nn.Softplus())])
        self.mlp_q.fc_out = nn.Linear(F, num_types)

    def _mlp(self, x):
        for fc in self.mlp_q.fcs:
            x = fc(x)
        return self.mlp_q.fc_out(x)

    def forward(self, data, one_hot):
        q = data["doping_charge"] if "doping_charge" in data else data.doping_charge
        q = torch.as_tensor(q).to(one_hot.dtype)
        q = q.view(1) if q.dim() == 0 else q
        q = q.view(-1, 1) if q.dim() == 1 else q
        batch = data.get("batch") if hasattr(data, "get") else getattr(data, "batch", None)
        if batch is not None and q.size(0) != one_hot.size(0):
            q = q[batch.view(-1)]
        elif q.size(0) != one_hot.size(0):
            q = q[:1].expand(one_hot.size(0), -1)
        q = q.clamp(self.charge_min, self.charge_max)
        diff = q - self.charge_centers.view(1, -1).to(one_hot.dtype)
        attrs = torch.exp(-self.charge_gamma.to(one_hot.dtype) * diff * diff)
        neutral = self.neutral_charge_attrs.to(one_hot.dtype).expand(attrs.size(0), -1)
        return one_hot + self._mlp(attrs) - self._mlp(neutral)


class HamGNNConvE3(nn.Module):
    """cfg: mapping/namespace with the reference's HamGNN_pre keys (hamgnn/models/hamgnn_conv.py:89-147)."""

    def __init__(self, cfg):
        super().__init__()
        c = cfg if isinstance(cfg, dict) else vars(cfg)
        c = c.get("HamGNN_pre", c)
        self.num_types = c["num_types"]
        self.irreps_edge_sh = Irreps(c["irreps_edge_sh"])
        self.cutoff, self.num_radial, self.num_layers = float(c["cutoff"]), c["num_radial"], c["num_layers"]
        self.sh_normalization = c.get("edge_sh_normalization", "component")
        self.sh_normalize = c.get("edge_sh_normalize", True)
        self.irreps_node_features = Irreps(c["irreps_node_features"])
        self.legacy_edge_update = c.get("legacy_edge_update", False)
        self.lite_mode = c.get("lite_mode", False)
        self.rbf_func = c.get("rbf_func", "bessel").lower()
        assert self.rbf_func in ("bessel", "gaussian") and not c.get("use_kan", False)
        assert not c.get("build_internal_graph", False)
        self.use_corr_prod = bool(c.get("use_corr_prod", False))
        mlp = list(c["radial_MLP"])
        attrs = Irreps([(self.num_types, (0, 1))])
        emb = Irreps([(self.num_radial, (0, 1))])
        D = self.irreps_node_features
        self.apply_charge_doping = bool(c.get("apply_charge_doping", False))
        if self.apply_charge_doping:                            # hamgnn_conv.py:147-153
            self.atomic_embedding = Embedding_block_q(self.num_types, int(c.get("num_charge_attr_feas", 8)))
        self.pair_embedding = PairInteractionEmbeddingBlock(attrs, self.irreps_edge_sh, emb, D, mlp, self.lite_mode)
        self.chemical_embedding = _Holder()
        self.chemical_embedding.linear = Linear(attrs, D)
        self.convolutions = nn.ModuleList()
        self.pair_interactions = nn.ModuleList()
        if self.use_corr_prod:                                  # hamgnn_conv.py:193-218
            from .mace_ref import CorrProductBlock
            self.corr_products = nn.ModuleList([CorrProductBlock(D, int(c["num_hidden_features"]), int(c["correlation"]), self.num_types, True)
                                                for _ in range(self.num_layers)])
        for i in range(self.num_layers):
            self.convolutions.append(ConvBlockE3(D, D, self.irreps_edge_sh, emb, mlp, self.lite_mode))
            skip = (i > 0) if self.legacy_edge_update else True
            self.pair_interactions.append(PairInteractionBlock(D, self.irreps_edge_sh, emb, D, mlp, skip, self.legacy_edge_update, self.lite_mode))

    def forward(self, data):
        dtype = self.chemical_embedding.linear.weight.dtype
        g = {"edge_index": data.edge_index}
        one_hot = torch.nn.functional.one_hot(data.z, self.num_types).to(dtype)
        if self.apply_charge_doping:
            one_hot = self.atomic_embedding(data, one_hot)
        g["node_attrs"] = g["node_features"] = one_hot
        sh, rbf, r = edge_geometry(data.pos.to(dtype), data.edge_index, data.nbr_shift.to(dtype), self.irreps_edge_sh, self.cutoff,
                                   self.num_radial, self.sh_normalize, self.sh_normalization, self.rbf_func)
        g["edge_attrs"], g["edge_embedding"] = sh, rbf
        self.pair_embedding(g)
        g["node_features"] = self.chemical_embedding.linear(g["node_features"])
        for i, (conv, pair) in enumerate(zip(self.convolutions, self.pair_interactions)):
            conv(g)
            if self.use_corr_prod:                              # hamgnn_conv.py:274-275
                g["node_features"] = self.corr_products[i](g["node_features"], g["node_attrs"])
            pair(g)
        return {"node_attr": g["node_features"], "edge_attr": g["edge_features"]}


# ---------------------------------------------------------------------------------------------- attention backbone


def soft_unit_step(x):
    """e3nn.math.soft_unit_step (e3nn 0.5.0 math/_soft_unit_step.py): exp(-1/x) for x > 0, else 0"""
    return torch.where(x > 0, torch.exp(-1.0 / torch.where(x > 0, x, torch.ones_like(x))), torch.zeros_like(x))


def edge_softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax(src, index) (PyG 2.x utils/_softmax.py; third-party, absent here -- its published algorithm):
    per group max subtracted, exp, divided by (group sum + 1e-16)"""
    mx = src.new_full((num_nodes,) + src.shape[1:], -float("inf"))
    mx = mx.scatter_reduce(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    return out / (scatter_sum(out, index, num_nodes) + 1e-16)[index]


def vector_to_heads(irreps_head, num_heads, x):
    """VectorToAttentionHeads (hamgnn/nn/attention_utils.py:17-48): every (mul * heads) x ir block viewed as [heads, mul * dim]"""
    out, i = [], 0
    for mul, ir in irreps_head:
        n = mul * num_heads * ir.dim
        out.append(x[:, i:i + n].reshape(x.shape[0], num_heads, -1))
        i += n
    assert i == x.shape[1], "irreps multiplicities must be divisible by num_heads"
    return torch.cat(out, 2)


def heads_to_vector_att(irreps_head, x):
    """AttentionHeadsToVector (hamgnn/nn/attention_utils.py:51-120)"""
    sizes = [mul * ir.dim for mul, ir in irreps_head]
    return torch.cat([t.reshape(x.shape[0], -1) for t in torch.split(x, sizes, dim=2)], 1)


class AttentionAggregation(nn.Module):
    """hamgnn/nn/attention.py:91-164; scale_irreps hamgnn/utils/irreps_utils.py:67-79"""

    def __init__(self, num_heads, irreps):
        super().__init__()
        self.num_heads = num_heads
        self.irreps_head = Irreps([(max(1, int(mul * (1 / num_heads))), ir) for mul, ir in Irreps(irreps)])

    def forward(self, key, value, query, edge_weight_cutoff, edge_index, num_nodes):
        key, value, query = (vector_to_heads(self.irreps_head, self.num_heads, t) for t in (key, value, query))
        dst = edge_index[1]
        w = (query * key).sum(-1)
        if edge_weight_cutoff is not None:
            w = edge_weight_cutoff[:, None] * w
        w = edge_softmax(w / math.sqrt(self.irreps_head.dim), dst, num_nodes).unsqueeze(-1)
        return heads_to_vector_att(self.irreps_head, scatter_sum(w * value, dst, num_nodes))


class AttentionBlockE3(nn.Module):
    """hamgnn/nn/attention.py:167-360.  As in the reference, BOTH key and query come from `linear_key` (`linear_query` is a parameter
    that the forward never reads, :339-340) and the soft cutoff carries the learnable `cutoff_func.cut_param`
    (hamgnn/utils/cutoff_functions.py:65-100)."""

    def __init__(self, irreps, irreps_edge_attrs, irreps_edge_embed, num_heads, max_radius, radial_MLP):
        super().__init__()
        self.register_buffer("max_radius", torch.tensor(float(max_radius)))
        self.cutoff_func = _Holder()
        self.cutoff_func.cut_param = nn.Parameter(torch.tensor(10.0))
        self.cutoff = float(max_radius)
        self.linear_up_src, self.linear_up_tar, self.linear_up_edge = Linear(irreps, irreps), Linear(irreps, irreps), Linear(irreps, irreps)
        self.residual = ResidualBlock(irreps, irreps)
        self.conv_tp_value = MessagePackBlock(irreps, irreps, irreps_edge_attrs, irreps, irreps_edge_embed, radial_MLP)
        self.linear_key, self.linear_query = Linear(irreps, irreps), Linear(irreps, irreps)
        self.attention = AttentionAggregation(num_heads, irreps)
        self.skip_linear = Linear(irreps, irreps)

    def forward(self, g):
        sender, receiver = g["edge_index"]
        x = g["node_features"]
        sc = self.skip_linear(x)
        key, query = self.linear_key(x)[sender], self.linear_key(x)[receiver]
        value = self.conv_tp_value(self.linear_up_src(x)[sender], self.linear_up_tar(x)[receiver], self.linear_up_edge(g["edge_features"]),
                                   g["edge_attrs"], g["edge_embedding"])
        cut = soft_unit_step(self.cutoff_func.cut_param * (1.0 - g["edge_lengths"] / self.cutoff))
        x = self.attention(key, value, query, cut, g["edge_index"], x.shape[0])
        g["node_features"] = self.residual(x) + sc
        return g["node_features"]


class HamGNNTransformer(nn.Module):
    """hamgnn/models/hamgnn_transformer.py:36-250: the HamGNNConvE3 pipeline with AttentionBlockE3 in place of ConvBlockE3 and a
    CorrProductBlock after every attention block (always on)."""

    def __init__(self, cfg):
        super().__init__()
        c = cfg if isinstance(cfg, dict) else vars(cfg)
        c = c.get("HamGNN_pre", c)
        self.num_types = c["num_types"]
        self.irreps_edge_sh = Irreps(c["irreps_edge_sh"])
        self.cutoff, self.num_radial, self.num_layers = float(c["cutoff"]), c["num_radial"], c["num_layers"]
        self.sh_normalization = c.get("edge_sh_normalization", "component")
        self.sh_normalize = c.get("edge_sh_normalize", True)
        self.irreps_node_features = Irreps(c["irreps_node_features"])
        self.rbf_func = c.get("rbf_func", "bessel").lower()
        assert self.rbf_func in ("bessel", "gaussian") and not c.get("use_kan", False) and not c.get("build_internal_graph", False)
        mlp = list(c["radial_MLP"])
        attrs = Irreps([(self.num_types, (0, 1))])
        emb = Irreps([(self.num_radial, (0, 1))])
        D = self.irreps_node_features
        self.apply_charge_doping = bool(c.get("apply_charge_doping", False))
        if self.apply_charge_doping:
            self.atomic_embedding = Embedding_block_q(self.num_types, int(c.get("num_charge_attr_feas", 8)))
        self.pair_embedding = PairInteractionEmbeddingBlock(attrs, self.irreps_edge_sh, emb, D, mlp)
        self.chemical_embedding = _Holder()
        self.chemical_embedding.linear = Linear(attrs, D)
        from .mace_ref import CorrProductBlock
        self.orb_transformers, self.corr_products, self.pair_interactions = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in range(self.num_layers):
            self.orb_transformers.append(AttentionBlockE3(D, self.irreps_edge_sh, emb, int(c["num_heads"]), self.cutoff, mlp))
            self.corr_products.append(CorrProductBlock(D, int(c["num_hidden_features"]), int(c["correlation"]), self.num_types, True))
            self.pair_interactions.append(PairInteractionBlock(D, self.irreps_edge_sh, emb, D, mlp, True, bool(c.get("legacy_edge_update", False))))

    def forward(self, data):
        dtype = self.chemical_embedding.linear.weight.dtype
        g = {"edge_index": data.edge_index}
        one_hot = torch.nn.functional.one_hot(data.z, self.num_types).to(dtype)
        if self.apply_charge_doping:
            one_hot = self.atomic_embedding(data, one_hot)
        g["node_attrs"] = g["node_features"] = one_hot
        sh, rbf, r = edge_geometry(data.pos.to(dtype), data.edge_index, data.nbr_shift.to(dtype), self.irreps_edge_sh, self.cutoff,
                                   self.num_radial, self.sh_normalize, self.sh_normalization, self.rbf_func)
        g["edge_attrs"], g["edge_embedding"], g["edge_lengths"] = sh, rbf, r
        self.pair_embedding(g)
        g["node_features"] = self.chemical_embedding.linear(g["node_features"])
        for att, corr, pair in zip(self.orb_transformers, self.corr_products, self.pair_interactions):
            att(g)
            g["node_features"] = corr(g["node_features"], g["node_attrs"])
            pair(g)
        return {"node_attr": g["node_features"], "edge_attr": g["edge_features"]}


# ---------------------------------------------------------------------------------------------- read-out head


def load_basis_tables():
    """Basis tables extracted (as data) from the reference by oracle/gen_golden.py -> tests/golden/basis_tables.json."""
    with open(os.path.join(_GOLDEN, "basis_tables.json")) as f:
        return json.load(f)


class HamLayer(nn.Module):
    def __init__(self, irreps_in, irreps_out, nonlinearity_type="gate"):
        super().__init__()
        self.residual_block = ResidualBlock(irreps_in, irreps_in, nonlinearity_type=nonlinearity_type)
        self.linear_transform = Linear(irreps_in, irreps_out)

    def forward(self, x):
        return self.linear_transform(self.residual_block(x))


def su2_required_irreps(row):
    """irreps of one complex half of the spinful decomposition (tensor_decomposition.py:40-88, 463-486): per (row shell, col
    shell) the L list of l_i x l_j, then for every L its coupling with the spin vector, L x 1 -> |L-1|..L+1; parity
    (-1)^(l_i+l_j) throughout."""
    out = Irreps([])
    for _, li in row:
        for _, lj in row:
            p = (-1) ** (li.l + lj.l)
            Ls = list(range(abs(li.l - lj.l), li.l + lj.l + 1))
            for L in Ls:
                out = out + Irrep(L, p)
            for L in Ls:
                for l2 in range(abs(L - 1), L + 2):
                    out = out + Irrep(l2, p)
    return out


class HamGNNPlusPlusOut(nn.Module):
    """Non-SOC branch and SOC/so3 branch of the reference head; ham_only=True; band/k-space code out of scope."""

    def __init__(self, irreps_in_node, irreps_in_edge, nao_max=19, ham_type="openmx", symmetrize=True, add_H0=True,
                 soc_switch=False, soc_basis="so3", add_H_nonsoc=False, zero_point_shift=False, ham_only=True, nonlinearity_type="gate"):
        super().__init__()
        self.nonlinearity_type = nonlinearity_type
        self.nao_max, self.ham_type = nao_max, ham_type.lower()
        self.symmetrize, self.add_H0, self.soc_switch, self.add_H_nonsoc = symmetrize, add_H0, soc_switch, add_H_nonsoc
        self.zero_point_shift = zero_point_shift
        self.soc_basis = soc_basis.lower() if self.ham_type == "openmx" or not soc_switch else "su2"
        assert self.soc_basis in ("so3", "su2")
        t = load_basis_tables()[f"{self.ham_type}_{nao_max}"]
        self.row = self.col = Irreps(t["row"])
        self.index_change = torch.tensor(t["index_change"]) if t["index_change"] is not None else None
        self.minus_index = torch.tensor(t["minus_index"]) if t.get("minus_index") is not None else None
        self.basis_def = {int(k): v for k, v in t["basis_def"].items()}
        self.num_valence = {int(k): v for k, v in t.get("num_valence", {}).items()}
        self.band_num_control = None
        irr = Irreps([])
        for _, li in self.row:
            for _, lj in self.col:
                for L in range(abs(li.l - lj.l), li.l + lj.l + 1):
                    irr = irr + Irrep(L, (-1) ** (li.l + lj.l))
        self.hamiltonian_irreps = irr
        self.ham_only = ham_only
        if not ham_only:                                        # hamgnn_output.py:247-256
            self.onsite_overlap_network = HamLayer(irreps_in_node, irr, nonlinearity_type)
            self.offsite_overlap_network = HamLayer(irreps_in_edge, irr, nonlinearity_type)
        if soc_switch and self.soc_basis == "su2":
            # E3TensorDecomposition(spinful=True).required_irreps_out (tensor_decomposition.py:463-527) is already the doubled
            # (re, im) list; the head doubles it once more (hamgnn_output.py:193,197) -- only copies 0 and 2 are ever read
            self.su2_required = su2_required_irreps(self.row)
            net = self.su2_required + self.su2_required
            net = net + net
            self.onsite_hamiltonian_network = HamLayer(irreps_in_node, net, nonlinearity_type)
            self.offsite_hamiltonian_network = HamLayer(irreps_in_edge, net, nonlinearity_type)
            return
        self.onsite_hamiltonian_network = HamLayer(irreps_in_node, irr, nonlinearity_type)
        self.offsite_hamiltonian_network = HamLayer(irreps_in_edge, irr, nonlinearity_type)
        if soc_switch:
            ksi = Irreps([(nao_max ** 2, (0, 1))])
            self.onsite_ksi_network = HamLayer(irreps_in_node, ksi, nonlinearity_type)
            self.offsite_ksi_network = HamLayer(irreps_in_edge, ksi, nonlinearity_type)

    # -- pieces
    def merge_tensor_components(self, coeff):
        Z = coeff.shape[0]
        H = coeff.new_zeros(Z, self.nao_max, self.nao_max)
        off, r0 = 0, 0
        for _, li in self.row:
            c0 = 0
            for _, lj in self.col:
                for L in range(abs(li.l - lj.l), li.l + lj.l + 1):
                    cg = math.sqrt(2 * L + 1) * e3.wigner_3j(li.l, lj.l, L, dtype=coeff.dtype)
                    H[:, r0:r0 + li.dim, c0:c0 + lj.dim] += torch.einsum("abm,zm->zab", cg, coeff[:, off:off + 2 * L + 1])
                    off += 2 * L + 1
                c0 += lj.dim
            r0 += li.dim
        return H.reshape(Z, -1)

    def reorder_matrix(self, M):
        M = M.reshape(-1, self.nao_max, self.nao_max)
        if self.index_change is not None:
            M = M[:, self.index_change[:, None], self.index_change[None, :]]
        if self.minus_index is not None:
            M = M.clone()
            M[:, self.minus_index, :] = -M[:, self.minus_index, :]
            M[:, :, self.minus_index] = -M[:, :, self.minus_index]
        return M.reshape(-1, self.nao_max ** 2)

    def _sym(self, M, inv=None, sign=1.0, dim=None):
        if not self.symmetrize:
            return M
        d = dim or self.nao_max
        A = M.reshape(-1, d, d)
        B = (A if inv is None else A[inv]).transpose(1, 2)
        return (0.5 * (A + sign * B)).reshape(-1, d * d)

    def orbital_mask(self, z, edge_index):
        table = torch.zeros(99, self.nao_max, dtype=torch.float64)
        for Z, idx in self.basis_def.items():
            table[Z, idx] = 1
        m = table[z]
        src, dst = edge_index
        on = (m[:, :, None] * m[:, None, :]).reshape(-1, self.nao_max ** 2)
        off = (m[src][:, :, None] * m[dst][:, None, :]).reshape(-1, self.nao_max ** 2)
        return on, off

    def interaction_masks(self, data, soc=False):
        """build_interaction_masks (hamgnn_output.py:2616-2665) / build_spin_orbit_interaction_masks (:2716-2783)."""
        on, off = self.orbital_mask(data.z, data.edge_index)
        on, off = on.bool(), off.bool()
        if not soc:
            return torch.cat([on, off], 0)
        n = self.nao_max
        big = lambda a: a.reshape(-1, n, n).repeat(1, 2, 2).reshape(-1, 4 * n * n)
        return self.cat_by_crystal(data, big(on), big(off))

    def ksi_average(self, ksi):
        """symmetrize_orbital_coefficients (hamgnn_output.py:2367-2431): mean over each (row shell, col shell) block."""
        K = ksi.reshape(-1, self.nao_max, self.nao_max).clone()
        r0 = 0
        for _, li in self.row:
            c0 = 0
            for _, lj in self.col:
                blk = K[:, r0:r0 + li.dim, c0:c0 + lj.dim]
                K[:, r0:r0 + li.dim, c0:c0 + lj.dim] = blk.mean(dim=(1, 2), keepdim=True).expand_as(blk)
                c0 += lj.dim
            r0 += li.dim
        return K.reshape(-1, self.nao_max ** 2)

    @staticmethod
    def global_inverse_edges(data):
        src = data.edge_index[0]
        if getattr(data, "batch", None) is None:
            return data.inv_edge_idx
        b = data.batch[src]
        counts = scatter_sum(torch.ones_like(src), b, int(data.batch.max()) + 1)
        offs = torch.cumsum(counts, 0) - counts
        return data.inv_edge_idx + offs[b]

    @staticmethod
    def cat_by_crystal(data, on, off):
        if getattr(data, "batch", None) is None or int(data.batch.max()) == 0:
            return torch.cat([on, off], 0)
        src = data.edge_index[0]
        nG = int(data.batch.max()) + 1
        nn_ = scatter_sum(torch.ones_like(data.batch), data.batch, nG).tolist()
        ne = scatter_sum(torch.ones_like(src), data.batch[src], nG).tolist()
        out = []
        for a, b in zip(torch.split(on, nn_), torch.split(off, ne)):
            out += [a, b]
        return torch.cat(out, 0)

    def soc_zero_point(self, data, Hr):
        """zero_point_shift of the non-collinear branches (hamgnn_output.py:3892-3913): spin-diagonal real blocks."""
        n = self.nao_max
        S = self.cat_by_crystal(data, data.Son, data.Soff).reshape(-1, n, n)
        H5 = Hr.reshape(-1, 2, n, 2, n).clone()
        R5 = self.cat_by_crystal(data, data.Hon, data.Hoff).reshape(-1, 2, n, 2, n)
        diff = (H5[:, 0, :, 0, :] + H5[:, 1, :, 1, :]) - (R5[:, 0, :, 0, :] + R5[:, 1, :, 1, :])
        sel = S > 1e-6
        dE = diff[sel].sum() / (2.0 * S[sel].sum())
        H5[:, 0, :, 0, :] -= dE * S
        H5[:, 1, :, 1, :] -= dE * S
        return H5.reshape(-1, 4 * n * n)

    def su2_get_H(self, net_out):
        """E3TensorDecomposition.get_H, spinful (tensor_decomposition.py:553-603): complex [Z, 4, nao, nao]."""
        half = net_out.shape[-1] // 2
        c = torch.complex(net_out[:, :half], net_out[:, half:])
        Z, n = c.shape[0], self.nao_max
        s2 = math.sqrt(2.0)
        spin = torch.tensor([[1, 0, 1, 0], [0, -1j, 0, 1], [0, 1j, 0, 1], [1, 0, -1, 0]], dtype=c.dtype) / s2
        H = c.new_zeros(Z, 4, n, n)
        off, r0 = 0, 0
        for _, li in self.row:
            c0 = 0
            for _, lj in self.col:
                Ls = list(range(abs(li.l - lj.l), li.l + lj.l + 1))
                m = li.dim * lj.dim
                cols = [c[:, off:off + m].unsqueeze(-1)]                                   # n = 0: the spin-scalar part
                o2 = off + m
                vec = []
                for L in Ls:                                                               # n = 1..3: (L x 1) -> L
                    Lp = list(range(abs(L - 1), L + 2))
                    w = torch.cat([e3.wigner_3j(L, 1, l2, dtype=net_out.dtype) for l2 in Lp], dim=-1).to(c.dtype)
                    d = sum(2 * l2 + 1 for l2 in Lp)
                    vec.append(torch.einsum("jkl,il->ijk", w, c[:, o2:o2 + d]))
                    o2 += d
                Hb = torch.cat([cols[0], torch.cat(vec, dim=-2)], dim=-1)                     # [Z, m, 4]
                wm = torch.cat([e3.wigner_3j(li.l, lj.l, L, dtype=net_out.dtype) for L in Ls], dim=-1).to(c.dtype)
                H[:, :, r0:r0 + li.dim, c0:c0 + lj.dim] += torch.einsum("imn,klm,jn->ijkl", Hb, wm, spin)
                off = o2
                c0 += lj.dim
            r0 += li.dim
        return H

    def forward_su2(self, data, rep, inv):
        """SOC / su2 branch (hamgnn_output.py:3146-3178, 3603-3625)."""
        n = self.nao_max

        def block(net, x, inv_):
            H = self.su2_get_H(net(x))                                                     # [Z, 4, n, n]
            H = self.reorder_matrix(H.reshape(-1, n * n)).reshape(-1, 2, 2, n, n).swapaxes(2, 3).reshape(-1, 2 * n, 2 * n)
            if self.symmetrize:
                H = 0.5 * (H + (H if inv_ is None else H[inv_]).conj().transpose(1, 2))
            return H.reshape(-1, 2, n, 2, n)

        on, off = block(self.onsite_hamiltonian_network, rep["node_attr"], None), block(self.offsite_hamiltonian_network, rep["edge_attr"], inv)
        m_on, m_off = self.orbital_mask(data.z, data.edge_index)
        m_on, m_off = m_on.reshape(-1, 1, n, 1, n).to(on.real.dtype), m_off.reshape(-1, 1, n, 1, n).to(on.real.dtype)
        on, off = (on * m_on).reshape(-1, 4 * n * n), (off * m_off).reshape(-1, 4 * n * n)
        on_r, on_i, off_r, off_i = on.real, on.imag, off.real, off.imag
        if self.add_H0:
            on_r, off_r = on_r + data.Hon0, off_r + data.Hoff0
            on_i, off_i = on_i + data.iHon0, off_i + data.iHoff0
        Hr, Hi = self.cat_by_crystal(data, on_r, off_r), self.cat_by_crystal(data, on_i, off_i)
        if self.zero_point_shift:
            Hr = self.soc_zero_point(data, Hr)
        return {"hamiltonian": torch.cat([Hr, Hi], 0), "hamiltonian_real": Hr, "hamiltonian_imag": Hi}

    def forward(self, data, rep):
        node_attr, edge_attr = rep["node_attr"], rep["edge_attr"]
        for Z in data.z.unique().tolist():
            if Z not in self.basis_def:
                raise ValueError(f"element Z={Z} missing from basis_def")
        inv = self.global_inverse_edges(data)
        if not self.ham_only:                                   # overlap matrices (hamgnn_output.py:2995-3019, 4009-4013)
            res = self._forward_hamiltonian(data, rep, inv)
            s_on = self._sym(self.reorder_matrix(self.merge_tensor_components(self.onsite_overlap_network(node_attr))))
            s_off = self._sym(self.reorder_matrix(self.merge_tensor_components(self.offsite_overlap_network(edge_attr))), inv)
            mo, mf = self.orbital_mask(data.z, data.edge_index)
            res["overlap"] = self.cat_by_crystal(data, s_on * mo.to(s_on.dtype), s_off * mf.to(s_on.dtype))
            return res
        return self._forward_hamiltonian(data, rep, inv)

    def _forward_hamiltonian(self, data, rep, inv):
        node_attr, edge_attr = rep["node_attr"], rep["edge_attr"]
        if self.soc_switch and self.soc_basis == "su2":
            return self.forward_su2(data, rep, inv)
        on = self._sym(self.reorder_matrix(self.merge_tensor_components(self.onsite_hamiltonian_network(node_attr))))
        off = self._sym(self.reorder_matrix(self.merge_tensor_components(self.offsite_hamiltonian_network(edge_attr))), inv)
        m_on, m_off = self.orbital_mask(data.z, data.edge_index)
        m_on, m_off = m_on.to(on.dtype), m_off.to(on.dtype)
        if not self.soc_switch:
            if self.add_H0:
                on, off = on + data.Hon0, off + data.Hoff0
            on, off = on * m_on, off * m_off
            H = self.cat_by_crystal(data, on, off)
            if self.zero_point_shift:
                S = self.cat_by_crystal(data, data.Son, data.Soff)
                Href = self.cat_by_crystal(data, data.Hon, data.Hoff)
                sel = S > 1e-6
                H = H - ((H - Href)[sel].sum() / S[sel].sum()) * S
            return {"hamiltonian": H, "H_on": on, "H_off": off}
        # ---- SOC / so3 (hamgnn_output.py:3026-3144, 3603-3625)
        n = self.nao_max
        Hon0, Hoff0 = data.Hon0, data.Hoff0
        if self.add_H_nonsoc:
            on, off = data.Hon_nonsoc, data.Hoff_nonsoc

            def zero_diag(h0):
                h0 = h0.reshape(-1, 2 * n, 2 * n).clone()
                h0[:, :n, :n] = 0
                h0[:, n:, n:] = 0
                return h0.reshape(-1, (2 * n) ** 2)
            Hon0, Hoff0 = zero_diag(Hon0), zero_diag(Hoff0)
        else:
            on, off = on * m_on, off * m_off
        ksi_on = self.ksi_average(self.onsite_ksi_network(node_attr))
        ksi_off = self.ksi_average(self.offsite_ksi_network(edge_attr))

        def spin_blocks(h, ksi, L, inv_):
            A = lambda k: self._sym(ksi * L[:, :, k], inv_, -1.0).reshape(-1, n, n)
            h3 = h.reshape(-1, n, n)
            real = h.new_zeros(h3.shape[0], 2 * n, 2 * n)
            imag = h.new_zeros(h3.shape[0], 2 * n, 2 * n)
            real[:, :n, :n] = h3
            real[:, n:, n:] = h3
            real[:, :n, n:] = A(1)
            real[:, n:, :n] = A(1)
            imag[:, :n, :n] = A(2)
            imag[:, n:, n:] = -A(2)
            imag[:, :n, n:] = A(0)
            imag[:, n:, :n] = -A(0)
            return real.reshape(-1, (2 * n) ** 2), imag.reshape(-1, (2 * n) ** 2)

        on_r, on_i = spin_blocks(on, ksi_on, data.Lon, None)
        off_r, off_i = spin_blocks(off, ksi_off, data.Loff, inv)
        if self.add_H0:
            on_r, off_r = on_r + Hon0, off_r + Hoff0
            on_i, off_i = on_i + data.iHon0, off_i + data.iHoff0
        Hr, Hi = self.cat_by_crystal(data, on_r, off_r), self.cat_by_crystal(data, on_i, off_i)
        if self.zero_point_shift:
            Hr = self.soc_zero_point(data, Hr)
        return {"hamiltonian": torch.cat([Hr, Hi], 0), "hamiltonian_real": Hr, "hamiltonian_imag": Hi}


# ---------------------------------------------------------------------------------------------- k-space step
def calculate_band_energies(self, onsite_hamiltonian, offsite_hamiltonian, data):
    """hamgnn/models/hamgnn_output.py:1675-1996 (export_reciprocal_values=False), same loop structure: per crystal, phase factors
    exp(2 pi i k . nbr_shift) (:1779-1788), padded [num_k, n, n, nao, nao] H(k) / S(k) with on-site blocks on the diagonal (:1813-1821) and
    off-site blocks accumulated at (source, target) (:1841-1862), orbital masked_select to the compact basis (:1897-1904), generalized
    eigenproblem through the Cholesky factor of S(k) (:1911-1928), band gap at the half-filled band (:1930-1936), optional band window.
    Returns (band_energy, wavefunction, band_gap, H_sym)."""
    nao = self.nao_max
    src, dst = data.edge_index
    k_vecs = data.k_vecs
    num_k = k_vecs.shape[1]
    mask_tab = torch.zeros(99, nao)
    for Z, idx in self.basis_def.items():
        mask_tab[Z][idx] = 1
    om = mask_tab[data.z]
    nval = torch.zeros(99)
    for Z, c in self.num_valence.items():
        nval[Z] = c
    node_counts = data.node_counts.tolist()
    n_off = np.cumsum([0] + node_counts)
    edge_counts = torch.bincount(data.batch[src], minlength=len(node_counts)).tolist()
    e_off = np.cumsum([0] + edge_counts)
    cdt = torch.complex128 if onsite_hamiltonian.dtype == torch.float64 else torch.complex64
    bands, waves, gaps, hsyms = [], [], [], []
    for c, n in enumerate(node_counts):
        sl_n, sl_e = slice(n_off[c], n_off[c + 1]), slice(e_off[c], e_off[c + 1])
        kp = k_vecs[c].to(onsite_hamiltonian.dtype)
        phase = torch.exp(2j * math.pi * (data.nbr_shift[sl_e][:, None, :].to(kp.dtype) * kp[None, :, :]).sum(-1))     # [e, nk]
        Hk = torch.zeros(num_k, n, n, nao, nao, dtype=cdt)
        Sk = torch.zeros(num_k, n, n, nao, nao, dtype=cdt)
        ar = torch.arange(n)
        Hk[:, ar, ar] += onsite_hamiltonian[sl_n].reshape(-1, nao, nao)[None].to(cdt)
        Sk[:, ar, ar] += data.Son[sl_n].reshape(-1, nao, nao)[None].to(cdt)
        si, ti = src[sl_e] - n_off[c], dst[sl_e] - n_off[c]
        Ho = offsite_hamiltonian[sl_e].reshape(-1, nao, nao).to(cdt)
        So = data.Soff[sl_e].reshape(-1, nao, nao).to(cdt)
        for k in range(num_k):
            ph = phase[:, k].to(cdt)[:, None, None]
            Hk[k] = torch.index_put(Hk[k], (si, ti), ph * Ho, accumulate=True)
            Sk[k] = torch.index_put(Sk[k], (si, ti), ph * So, accumulate=True)
        Hk = Hk.swapaxes(-2, -3).reshape(num_k, n * nao, n * nao)
        Sk = Sk.swapaxes(-2, -3).reshape(num_k, n * nao, n * nao)
        m = om[sl_n].reshape(-1)
        keep = (m[:, None] * m[None, :] > 0)[None].expand(num_k, -1, -1)
        norb = int(m.sum())
        Hk = torch.masked_select(Hk, keep).reshape(num_k, norb, norb)
        Sk = torch.masked_select(Sk, keep).reshape(num_k, norb, norb)
        L = torch.linalg.cholesky(Sk)
        LH = L.conj().transpose(-1, -2)
        Linv, LHinv = torch.linalg.inv(L), torch.linalg.inv(LH)
        Ht = torch.bmm(torch.bmm(Linv, Hk), LHinv)
        ev, evec = torch.linalg.eigh(Ht)
        evec = torch.einsum("ijk,ika->iaj", LHinv, evec)
        half = math.ceil(float(nval[data.z[sl_n]].sum()) / 2)
        gaps.append((ev[:, half].min() - ev[:, half - 1].max()).reshape(1))
        bnc = self.band_num_control
        if bnc is not None:
            if isinstance(bnc, dict):
                nb = int(sum(bnc[int(zz)] for zz in data.z[sl_n].tolist()))
                ev, evec = ev[:, :nb], evec[:, :nb, :]
            else:
                win = max(1, int(bnc * half)) if isinstance(bnc, float) else min(bnc, half)
                ev, evec = ev[:, half - win:half + win], evec[:, half - win:half + win, :]
        bands.append(ev.transpose(-1, -2))
        waves.append(evec.reshape(-1))
        hsyms.append(Ht.reshape(-1))
    return torch.cat(bands, 0), torch.cat(waves, 0), torch.cat(gaps, 0), torch.cat(hsyms, 0)


HamGNNPlusPlusOut.calculate_band_energies = calculate_band_energies


def calculate_band_energies_with_spin_orbit_coupling(self, real_onsite, imag_onsite, real_offsite, imag_offsite, data):
    """hamgnn/models/hamgnn_output.py:1998-2286: spinor blocks [., (2 nao)^2] -> (band_energy, wavefunction).  Same steps: per crystal the
    overlap S(k) = on-site + sum over edges of exp(2 pi i k . nbr_shift) S_off (the reference sums per unique cell shift first, :2131-2150:
    the same sum), masked to the atoms' orbitals, kron(1_2, S(k)) (:2165-2167); the four spin blocks of H(k) likewise from
    real + i imag (:2170-2233); Cholesky / eigh (:2236-2252); band window (:2254-2263: dict -> leading bands, int -> +- around the number
    of valence electrons)."""
    nao = self.nao_max
    src, dst = data.edge_index
    k_vecs = data.k_vecs
    num_k = k_vecs.shape[1]
    mask_tab = torch.zeros(99, nao)
    for Z, idx in self.basis_def.items():
        mask_tab[Z][idx] = 1
    om = mask_tab[data.z]
    nval = torch.zeros(99)
    for Z, c in self.num_valence.items():
        nval[Z] = c
    node_counts = data.node_counts.tolist()
    n_off = np.cumsum([0] + node_counts)
    edge_counts = torch.bincount(data.batch[src], minlength=len(node_counts)).tolist()
    e_off = np.cumsum([0] + edge_counts)
    rdt = real_onsite.dtype
    cdt = torch.complex128 if rdt == torch.float64 else torch.complex64
    spin = lambda t: t.reshape(-1, 2, nao, 2, nao)
    bands, waves = [], []
    for c, n in enumerate(node_counts):
        sl_n, sl_e = slice(n_off[c], n_off[c + 1]), slice(e_off[c], e_off[c + 1])
        kp = k_vecs[c].to(rdt)
        phase = torch.exp(2j * math.pi * (data.nbr_shift[sl_e][:, None, :].to(rdt) * kp[None, :, :]).sum(-1)).to(cdt)        # [e, nk]
        si, ti = src[sl_e] - n_off[c], dst[sl_e] - n_off[c]
        m = om[sl_n].reshape(-1)
        keep = m[:, None] * m[None, :] > 0
        norb = int(m.sum())
        ar = torch.arange(n)

        def to_k(on_blk, off_blk):                             # [n, nao, nao], [e, nao, nao] (complex) -> [nk, norb, norb]
            out = torch.zeros(num_k, n, n, nao, nao, dtype=cdt)
            out[:, ar, ar] += on_blk[None].to(cdt)
            for k in range(num_k):
                out[k] = torch.index_put(out[k], (si, ti), phase[:, k][:, None, None] * off_blk.to(cdt), accumulate=True)
            out = out.swapaxes(2, 3).reshape(num_k, n * nao, n * nao)
            return out[:, keep].reshape(num_k, norb, norb)
        Sk = to_k(data.Son[sl_n].reshape(-1, nao, nao), data.Soff[sl_e].reshape(-1, nao, nao))
        Ssoc = torch.kron(torch.eye(2, dtype=cdt), Sk)
        ron, ion, roff, ioff = spin(real_onsite[sl_n]), spin(imag_onsite[sl_n]), spin(real_offsite[sl_e]), spin(imag_offsite[sl_e])
        blocks = [[to_k(ron[:, a, :, b, :] + 1j * ion[:, a, :, b, :], roff[:, a, :, b, :] + 1j * ioff[:, a, :, b, :]) for b in (0, 1)] for a in (0, 1)]
        Hk = torch.cat([torch.cat(blocks[0], -1), torch.cat(blocks[1], -1)], -2)
        L = torch.linalg.cholesky(Ssoc)
        LH = L.conj().transpose(-1, -2)
        Linv, LHinv = torch.linalg.inv(L), torch.linalg.inv(LH)
        ev, evec = torch.linalg.eigh(torch.bmm(torch.bmm(Linv, Hk), LHinv))
        evec = torch.bmm(LHinv, evec)
        bnc = self.band_num_control
        if bnc is not None:
            if isinstance(bnc, dict):
                nb = int(sum(bnc[int(zz)] for zz in data.z[sl_n].tolist()))
                ev, evec = ev[:, :nb], evec[:, :nb, :]
            else:
                nv = int(nval[data.z[sl_n]].sum())
                ev, evec = ev[:, nv - bnc:nv + bnc], evec[:, nv - bnc:nv + bnc, :]
        bands.append(ev.transpose(-1, -2))
        waves.append(evec.reshape(-1))
    return torch.cat(bands, 0), torch.cat(waves, 0)


HamGNNPlusPlusOut.calculate_band_energies_with_spin_orbit_coupling = calculate_band_energies_with_spin_orbit_coupling

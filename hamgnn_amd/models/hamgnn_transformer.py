"""HamGNNTransformer -- MI355X drop-in for the reference's attention backbone (hamgnn/models/hamgnn_transformer.py:36-250): the
HamGNNConvE3 pipeline with an AttentionBlockE3 (hamgnn/nn/attention.py:167-360) in place of every ConvBlockE3 and a CorrProductBlock
after it (always on).  Same config keys (+ num_heads), parameter names (`orb_transformers.{i}.*`, `corr_products.{i}.*`,
`pair_interactions.{i}.*`) and result dict as the reference.

On the device an attention block is: five node/edge-row o3.Linears (fused Linear programs), the value MessagePackBlock on the
input-stationary MFMA kernel (the same K2 launch as a ConvBlockE3's, un-rotated in its epilogue), `hg_attn_logits` +
`hg_attn_aggregate` (per-head soft-max over a node's incoming edges and the weighted sum of the value rows in one pass over them,
csrc/attention.hip) and the Gate ResidualBlock with the skip row added in its last Linear's epilogue."""
from __future__ import annotations

import torch
from torch import nn

from .. import nn as hnn
from .hamgnn_conv import _BackboneBase, _cfg_get


class HamGNNTransformer(_BackboneBase):
    def __init__(self, config):
        super().__init__()
        self.use_corr_prod = True                              # hamgnn_transformer.py:139-147: unconditional
        g = self._init_common(config)
        if self.lite_mode:
            raise NotImplementedError("HamGNNTransformer has no lite_mode in the reference (hamgnn_transformer.py:126-161)")
        D, sh, R, mlp = self.irreps_node_features, self.irreps_edge_sh, self.num_radial, self.radial_MLP
        self.num_heads = int(g("num_heads"))
        self.orb_transformers, self.corr_products, self.pair_interactions = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in range(self.num_layers):
            self.orb_transformers.append(hnn.AttentionBlockE3(D, sh, R, self.num_heads, self.cutoff, mlp))
            self.corr_products.append(hnn.CorrProductBlock(D, int(g("num_hidden_features")), int(g("correlation")), self.num_types, True))
            self.pair_interactions.append(hnn.PairInteractionBlock(D, sh, R, mlp, True, self.legacy_edge_update, False))
        # structural zeros of the first layer (see HamGNNConvE3._mark_structural_zeros): the value block reads linear_up_src / _tar (node) and linear_up_edge (f),
        # same-irrep o3.Linears that keep zero blocks zero (attention.py:339-352); the first pair block reads full node rows and the embedding's edge rows
        zero_node, zero_edge = self._structural_zero_sets()
        self.orb_transformers[0].conv_tp_value.set_structural_zeros(node=zero_node, edge=zero_edge)
        self.pair_interactions[0].conv_tp.set_structural_zeros(node=(), edge=zero_edge)

    def compile(self, device):
        dev = torch.device(device)
        for a, c, p in zip(self.orb_transformers, self.corr_products, self.pair_interactions):
            a.compile(dev)
            c.compile(dev)
            p.compile(dev)
        self._compile_common(dev)
        return self

    def refresh_weights(self):
        """after an optimiser step (hamgnn_amd.training): as HamGNNConvE3.refresh_weights -- message blocks repacked on the device, the
        small tables on the host"""
        dev = self._compiled_for
        if dev is None:
            return
        for att, corr, pair in zip(self.orb_transformers, self.corr_products, self.pair_interactions):
            att.refresh(dev)
            corr.compile(dev)
            pair.refresh(dev)
        self._compile_common(dev)

    def forward(self, data, save_for_backward: bool = False):
        """save_for_backward: keep the layer inputs on the result (`_tape`) for `backward`"""
        z, topo, geo, node, f = self._embed(data)
        from .. import parallel
        rowptr, perm = topo.receiver_csr()
        tape = [] if save_for_backward else None
        last = self.pair_interactions[-1]
        has_dead, skip_dead = self._dead_plan(tape)             # (unread irreps of the last pair block: _BackboneBase.declare_consumer)
        for att, corr, pair in zip(self.orb_transformers, self.corr_products, self.pair_interactions):
            if tape is not None:
                tape.append(dict(node_in=node, f_in=f))
            node = att.run(node, f, geo, self._rot_tab, rowptr, perm, data, structural_zeros=True)        # AttentionBlockE3.forward (attention.py:315-360)
            if tape is not None:
                tape[-1]["node_att"] = node
            node = corr(node, z, self._last_delta)                                  # CorrProductBlock.forward (interaction_blocks.py:234-260)
            if tape is not None:
                tape[-1]["node_out"] = node
            f_in, f = f, self._run_pair(pair, node, f, geo, reduced=(pair is not last) or skip_dead or not has_dead)
        rep = self._representation(node, f, geo, (lambda: self._run_pair(last, node, f_in, geo, reduced=False)) if skip_dead else None)
        if tape is not None:
            rep["_tape"] = tape
            if self._last_delta is not None:
                rep["_charge_delta"] = self._last_delta
        return rep

    # ------------------------------------------------------------------------------------------------------------ backward (SURVEY 8f-3 x 8f-4)
    def backward(self, data, rep, g_node, g_edge_rot, chunk: int = 65536):
        """as HamGNNConvE3.backward: gradients of every backbone parameter for the gradients of the representation (planar node rows,
        edge-frame edge rows); per layer, last to first: PairInteractionBlock -> CorrProductBlock -> AttentionBlockE3; then the embeddings."""
        from ..topo import get_topology
        tape, geo = rep["_tape"], rep["_geometry"]
        z = data.z.contiguous()
        topo = get_topology(data)
        grads = {}
        g_node, g_f = g_node.contiguous(), g_edge_rot.contiguous()
        for li in reversed(range(self.num_layers)):
            att, corr, pair, t = self.orb_transformers[li], self.corr_products[li], self.pair_interactions[li], tape[li]
            g_node, g_f = self._backward_pair(li, pair, t["node_out"], t["f_in"], geo, topo, g_node, g_f, grads, chunk, data)
            g_node, g_cp = corr.backward(t["node_att"], z, g_node, delta=rep.get("_charge_delta"))
            self._add_g_delta(rep, g_cp.pop("_g_delta", None))
            grads.update({f"corr_products.{li}." + k: v for k, v in g_cp.items()})
            g_node, g_f_att, g_at = att.backward(t["node_in"], t["f_in"], geo, self._rot_tab, topo, g_node, chunk=chunk, data=data)
            grads.update({f"orb_transformers.{li}." + k: v for k, v in g_at.items()})
            g_f = g_f + g_f_att
        self._backward_embeddings(data, rep, geo, g_node, g_f, grads, chunk)
        return grads

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

    def compile(self, device):
        dev = torch.device(device)
        for a, c, p in zip(self.orb_transformers, self.corr_products, self.pair_interactions):
            a.compile(dev)
            c.compile(dev)
            p.compile(dev)
        self._compile_common(dev)
        return self

    def forward(self, data):
        z, topo, geo, node, f = self._embed(data)
        shard = data.get("_hg_shard") if hasattr(data, "get") else None
        if shard is not None and shard[1] != 1:
            raise NotImplementedError("HamGNNTransformer on an edge-sharded graph: the per-node soft-max needs a max / sum exchange "
                                      "between the ranks that is not built (single-GPU only)")
        rowptr, perm = topo.receiver_csr()
        for att, corr, pair in zip(self.orb_transformers, self.corr_products, self.pair_interactions):
            node = att.run(node, f, geo, self._rot_tab, rowptr, perm)              # AttentionBlockE3.forward (attention.py:315-360)
            node = corr(node, z)                                                    # CorrProductBlock.forward (interaction_blocks.py:234-260)
            f = self._run_pair(pair, node, f, geo)
        return self._representation(node, f, geo)

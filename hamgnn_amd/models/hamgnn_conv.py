"""HamGNNConvE3 -- MI355X drop-in for the reference backbone (hamgnn/models/hamgnn_conv.py:88-284): same constructor
config keys, same parameter names, same return dict {node_attr [N,D], edge_attr [E,D]} (e3nn layout, global frame).
Underneath: planar feature rows, per-edge Wigner frames and the fused MFMA edge kernel (csrc/tp_fused.hip)."""
from __future__ import annotations

import math

import numpy as np
import os

import torch
from torch import nn

from .. import nn as hnn
from .. import ops
from .. import parallel
from .. import plan as P
from ..so3 import Irreps
from ..topo import get_topology, gget


class Representation(dict):
    """EasyDict-like result: key and attribute access (reference returns an EasyDict, hamgnn_conv.py:278-284).
    `node_attr` / `edge_attr` (e3nn layout, global frame) are produced ON FIRST ACCESS: the MI355X head reads the planar / edge-frame
    tensors directly (`_node_planar`, `_edge_planar_rot`), so a forward that never looks at them skips the layout conversion and the
    un-rotation of all E edge rows (4.5 ms per forward at 0.82 M edges)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        dict.__setitem__(self, "_lazy", {})

    def set_lazy(self, key, thunk):
        dict.__getitem__(self, "_lazy")[key] = thunk

    def __missing__(self, key):
        lazy = dict.__getitem__(self, "_lazy")
        if key in lazy:
            v = lazy.pop(key)()
            dict.__setitem__(self, key, v)
            return v
        raise KeyError(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in dict.__getitem__(self, "_lazy")

    def keys(self):
        return [k for k in dict.keys(self) if k != "_lazy"] + list(dict.__getitem__(self, "_lazy"))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _cfg_get(c, k, default=None):
    if isinstance(c, dict):
        return c.get(k, default)
    return getattr(c, k, default)


class _BackboneBase(nn.Module):
    """What HamGNNConvE3 and HamGNNTransformer share (hamgnn_conv.py:89-190 == hamgnn_transformer.py:37-112): config keys, atomic /
    pair / chemical embedding, per-edge geometry, and the lazily converted result dict."""

    def _structural_zero_sets(self):
        """(node irreps, edge irreps) that are STRUCTURALLY zero in the rows the first layer reads: the chemical embedding is an o3.Linear from
        `num_types x 0e` (toolbox/nequip/nn/_atomwise.py:55-57; charge doping adds to the same 0e attributes) -- only 0e blocks of the node rows can be
        non-zero -- and the pair embedding is 0e (x) Y^l followed by same-irrep Linears (nn/embeddings.py:310-337) -- only the irreps of the spherical
        harmonics can be non-zero in the edge rows"""
        from ..so3 import Irreps
        irr = Irreps(self.irreps_node_features)
        sh = {(l, p) for _, l, p in Irreps(self.irreps_edge_sh)}
        return (tuple(i for i, (_, l, p) in enumerate(irr) if (l, p) != (0, 1)), tuple(i for i, (_, l, p) in enumerate(irr) if (l, p) not in sh))

    def _init_common(self, config):
        c = _cfg_get(config, "HamGNN_pre", config)
        g = lambda k, d=None: _cfg_get(c, k, d)
        self.num_types = g("num_types")
        self.irreps_edge_sh = Irreps(g("irreps_edge_sh"))
        self.cutoff = float(g("cutoff"))
        self.num_radial = g("num_radial")
        self.num_layers = g("num_layers")
        self.irreps_node_features = Irreps(g("irreps_node_features"))
        self.radial_MLP = list(g("radial_MLP"))
        self.legacy_edge_update = bool(g("legacy_edge_update", False))
        self.rbf_func = str(g("rbf_func", "bessel")).lower()
        if self.rbf_func in ("exp-gaussian", "exp-bernstein", "bernstein"):
            # these reference bases hold float64 buffers and return a float64 edge embedding (utils/basis_functions.py:16-105): they only
            # run under `precision: 64`, which is not built here
            raise NotImplementedError(f"rbf_func={self.rbf_func!r} needs the reference's fp64 mode (precision: 64), which is not built")
        if self.rbf_func not in ("bessel", "gaussian"):
            raise ValueError(f"Unsupported radial basis function: {g('rbf_func')}")      # hamgnn_conv.py:139-141
        self.lite_mode = bool(g("lite_mode", False))
        for k in ("use_kan", "build_internal_graph"):
            if g(k, False):
                raise NotImplementedError(f"HamGNN_pre.{k}=True is outside the MI355X hot-path scope of this round (SURVEY 8f)")
        # use_gradient_checkpointing (hamgnn_conv.py:40-85, 236-246: torch.utils.checkpoint around every layer) is accepted and has nothing to
        # switch: the backward here keeps ONLY the layer inputs (forward(save_for_backward=True): node rows, edge rows, aggregates) and
        # re-evaluates every intermediate inside the block backwards -- the memory profile the reference's flag buys
        self.use_gradient_checkpointing = bool(g("use_gradient_checkpointing", False))
        self.apply_charge_doping = bool(g("apply_charge_doping", False))                 # hamgnn_conv.py:147-153
        if self.apply_charge_doping:
            self.atomic_embedding = hnn.ChargeEmbedding(self.num_types, int(g("num_charge_attr_feas", 8)))
        if g("edge_sh_normalization", "component") != "component" or not g("edge_sh_normalize", True):
            raise NotImplementedError("only component-normalised, normalised edge SH are supported")
        for _, l, p in self.irreps_edge_sh:
            assert p == (-1) ** l
        D, sh, R, mlp = self.irreps_node_features, self.irreps_edge_sh, self.num_radial, self.radial_MLP
        self.lmax = max(D.lmax, sh.lmax)
        self.pair_embedding = hnn.PairInteractionEmbeddingBlock(self.num_types, sh, D, R, mlp, self.lite_mode)
        self.chemical_embedding = nn.Module()
        self.chemical_embedding.linear = hnn.E3Linear(Irreps([(self.num_types, 0, 1)]), D)
        self.layout = P.PlanarLayout(D)
        self._compiled_for = None
        return g

    def _compile_common(self, dev):
        self.pair_embedding.compile(dev)
        lay = self.layout
        if getattr(self, "_struct_for", None) != dev:          # structural tables: once per device (not per repack)
            self._imap = torch.from_numpy(lay.index_map().astype(np.int32)).to(dev)
            self._rot_tab = torch.from_numpy(P.rotate_table(lay)).to(dev)
            self._jtab = torch.from_numpy(P.wigner_jtab(self.lmax)).to(dev)
            self._struct_for = dev
        # chemical embedding = row look-up of o3.Linear(num_types x 0e -> D) applied to one-hot rows (planar table)
        T = self.num_types
        W = self.chemical_embedding.linear.weight.detach().cpu().double().numpy()
        table = np.zeros((T, lay.dim))
        off = 0
        for k, (mk, lk, pk) in enumerate(self.irreps_node_features):
            if (lk, pk) == (0, 1):
                table[:, lay.off[k]:lay.off[k] + mk] = W[off:off + T * mk].reshape(T, mk) / math.sqrt(T)
                off += T * mk
        assert off == W.size
        self._chem = torch.from_numpy(table.astype(np.float32)).to(dev)
        self._compiled_for = dev

    def _embed(self, data):
        """-> (z, topology, geometry, node [N, Dp] planar, f [E, Dp] planar in the edge frame)"""
        ops.require_fp32(self, data)                           # `precision: 64` raises instead of returning fp32-accurate rows
        dev = data.pos.device
        if getattr(self, "_pending_refresh", False):           # an optimiser stepped since the last forward (training.weights_changed)
            self._pending_refresh = False
            self.refresh_weights()
        if self._compiled_for != dev:
            self.compile(dev)
        N = data.z.shape[0]
        z = data.z.contiguous()
        topo = get_topology(data)                              # index plumbing + validation: once per graph object (host-syncs)
        topo.check_num_types(self.num_types)                   # z >= num_types would index past the embedding tables on the device
        geo = ops.Geometry(data.pos, data.edge_index, data.nbr_shift, self.cutoff, self.num_radial, self.lmax, self._jtab, self.rbf_func)
        # hidden activations of ALL radial weight generators of this forward (embedding + two per message block) in one launch
        ops.prefill_radial_hidden(geo, self._radial_generators(), float(P.ACT_CONSTS[P.ACT_SILU]))
        Dp = self.layout.dim
        delta = None
        if self.apply_charge_doping:                           # node_attrs = one_hot(z) + delta (toolbox/nequip/nn/embedding/_embedding_block.py:124-131)
            q = gget(data, "doping_charge")
            if q is None:
                raise ValueError("apply_charge_doping=True needs data.doping_charge (scalar, one value per crystal, or one per atom)")
            delta = self.atomic_embedding.to(dev).delta(q, gget(data, "batch"), N, dev)
        self._last_delta = delta
        f = self.pair_embedding.run(z, geo, delta)                                   # [E, Dp] edge-aligned frame
        if delta is None:
            node = ops.embed_lookup(self._chem, None, z, None, None, N, Dp, Dp)      # [N, Dp]
        else:
            node = (self._chem[z] + delta @ self._chem).contiguous()                # per-atom rows of the same table
        return z, topo, geo, node, f

    def _radial_generators(self):
        gens = [self.pair_embedding._h]
        for m in self.modules():
            if isinstance(m, hnn.MessagePackBlock) and m._dp is not None:
                gens += [g for g in (m._hn, m._he) if g is not None]
        return gens

    def _representation(self, node, f, geo, full_edge_rows=None):
        """full_edge_rows: None, or -- when `f` holds only the irreps the declared consumer reads (declare_consumer) -- a thunk that evaluates the COMPLETE
        edge rows; the public `edge_attr` goes through it, so anything but the declared head sees exactly what the reference returns"""
        rep = Representation()
        imap, rot_tab = self._imap, self._rot_tab
        rep.set_lazy("node_attr", lambda: ops.from_planar(node, imap))
        if full_edge_rows is None:
            rep.set_lazy("edge_attr", lambda: ops.from_planar(ops.rotate_gather(f, None, geo, rot_tab, transpose=True), imap))
        else:
            # (the two thunks share the complete rows through `once`, NOT through `rep`: a closure over the representation would be a reference cycle, and a
            # cycle keeps ~9 GB of rows per forward at 0.82 M edges alive until Python's cycle collector runs -- measured as one 500 ms step in five)
            once = []

            def complete():
                if not once:
                    once.append(full_edge_rows())
                return once[0]
            rep["_edge_alive"] = self._edge_alive
            rep.set_lazy("_edge_planar_rot_full", complete)
            rep.set_lazy("edge_attr", lambda: ops.from_planar(ops.rotate_gather(complete(), None, geo, rot_tab, transpose=True), imap))
        # extras for the MI355X head: skip the layout/frame round trip
        rep["_node_planar"], rep["_edge_planar_rot"], rep["_geometry"] = node, f, geo
        return rep

    # ---- unread irreps of the last PairInteractionBlock (r5)
    _edge_alive = None

    def declare_consumer(self, head):
        """Tell the backbone that `head` is the ONLY reader of the representation it returns (what `Model(representation, output)` wires: Model.py:459-465 of
        the reference passes the representation to the output module and nowhere else).  If the head can say which (l, p) classes of the edge rows it reads
        (HamGNNPlusPlusOut.edge_irreps_read), the LAST PairInteractionBlock (of either backbone) stops computing the others: its reduced program drops their super-paths
        (plan.build_message_pack_program dead_out; set-A with nao_max 19: 0o, 4o, 5o, 5e, 6e = 15 % of that launch's MFMAs).  What the reference API promises
        stays true: `rep["edge_attr"]` (and a head that reads more than the declared one) gets the complete rows -- the block's inputs are kept on the
        representation and the complete program runs on first access.  Training forwards (save_for_backward) always run the complete program.
        HG_DEAD_OUT=0 disables.  Returns the list of dropped irreps (indices into irreps_node_features)."""
        self._edge_alive = None
        pairs = getattr(self, "pair_interactions", None)
        if pairs is None or self.lite_mode:
            return []
        last = pairs[-1]
        if (not hasattr(head, "edge_irreps_read") or os.environ.get("HG_DEAD_OUT", "1") == "0"
                or not (last.use_skip_connections or not last.legacy_edge_update)):      # (a legacy single-layer backbone: the block is not evaluated at all)
            if getattr(last.conv_tp, "_dead", ()):              # a consumer that says nothing after one that did: back to the complete program
                last.conv_tp.set_dead_outputs(())
                self._compiled_for = None
            return []
        need = head.edge_irreps_read()
        dead = [k for k, (m, l, p) in enumerate(self.irreps_node_features) if (int(l), int(p)) not in need]
        last.conv_tp.set_dead_outputs(dead)
        if dead:
            self._edge_alive = frozenset((int(l), int(p)) for k, (m, l, p) in enumerate(self.irreps_node_features) if k not in dead)
        self._compiled_for = None                                # the reduced program is built at the next compile()
        return dead

    def _dead_plan(self, tape):
        """(the last pair block has declared dead outputs, this forward may skip them).  Not while training: a block with declared dead outputs then runs
        its complete program."""
        has_dead = "dead_out" in getattr(self.pair_interactions[-1].conv_tp, "_zkw_compiled", {})     # (as compiled: the reduced program that would run)
        return has_dead, has_dead and tape is None and self._edge_alive is not None

    def _run_pair(self, pair, node, f, geo, reduced=True, merged_up=False):
        """PairInteractionBlock.forward (interaction_blocks.py:130-164).  reduced: the block may run its reduced program (structurally zero inputs of a first
        layer, unread outputs of a last layer whose consumer was declared)"""
        if pair.use_skip_connections or not pair.legacy_edge_update:               # legacy layer-0: edge features kept (:154-156)
            # (structural_zeros: inside a backbone's forward the rows are what set_structural_zeros was told about -- a first-layer block runs its reduced program)
            up_s, up_t = pair.linear_up_both(node) if (merged_up and pair.conv_tp._dp.sched is not None) else (pair.linear_up_src(node), pair.linear_up_tar(node))
            mix = pair.conv_tp.run_nodes(up_s, up_t, f, geo, self._rot_tab, structural_zeros=reduced)   # edge frame (+ fused skip linear)
            if self.lite_mode and pair.use_skip_connections:
                mix = pair.skip_linear(f, res=[mix])
            f = mix
        return f


    # ---- backward pieces shared by the two backbones
    def _backward_pair(self, li, pair, node_out, f_in, geo, topo, g_node, g_f, grads, chunk, data=None):
        """PairInteractionBlock (interaction_blocks.py:130-164): f_out = MP(up_src(node_out)[src], up_tar(node_out)[dst], f_in) + skip(f_in).
        g_node: gradient of node_out so far, g_f: gradient of f_out (edge frame).  Returns (g_node, gradient of f_in); parameter
        gradients go into `grads`."""
        N = node_out.shape[0]
        rp_r, pm_r = topo.receiver_csr()
        rp_s, pm_s = topo.sender_csr()
        pre = f"pair_interactions.{li}."
        if pair.use_skip_connections or not pair.legacy_edge_update:
            up_s, up_t = pair.linear_up_src(node_out), pair.linear_up_tar(node_out)
            # (structural_zeros: as in the forward -- a first-layer block skips the paths that read structurally zero input irreps: zero weight gradients, unread data gradients)
            gs, gd, ge, g_tp = pair.conv_tp.backward(up_s, up_t, f_in, geo, self._rot_tab, g_f, out_is_global=False, chunk=chunk, structural_zeros=True)
            grads.update({pre + "conv_tp." + k: v for k, v in g_tp.items()})
            g_up_s = ops.segment_sum(gs, rp_s, pm_s, N)
            g_up_t = ops.segment_sum(gd, rp_r, pm_r, N)
            if data is not None:                               # edge-sharded: sums over THIS rank's edges -> sums over all edges (RCCL)
                parallel.allreduce_nodes(g_up_s, data)
                parallel.allreduce_nodes(g_up_t, data)
            grads[pre + "linear_up_src.weight"] = pair.linear_up_src.weight_grad(node_out, g_up_s)
            grads[pre + "linear_up_tar.weight"] = pair.linear_up_tar.weight_grad(node_out, g_up_t)
            g_node = g_node + pair.linear_up_src.backward_data(g_up_s) + pair.linear_up_tar.backward_data(g_up_t)
            if pair.use_skip_connections:
                grads[pre + "skip_linear.weight"] = pair.skip_linear.weight_grad(f_in, g_f)
                ge = ge + pair.skip_linear.backward_data(g_f)
            g_f = ge
        else:                                                  # legacy layer 0: the block is not evaluated, its parameters get zeros
            for k, p_ in pair.named_parameters():
                grads[pre + k] = torch.zeros_like(p_).reshape(-1)
        return g_node, g_f

    @staticmethod
    def _add_g_delta(rep, g):
        """gradient with respect to the charge-doping correction from a consumer other than the embeddings (the CorrProductBlocks)"""
        if g is not None:
            rep["_g_delta_extra"] = g if rep.get("_g_delta_extra") is None else rep["_g_delta_extra"] + g

    def _backward_embeddings(self, data, rep, geo, g_node, g_f, grads, chunk):
        """edge rows from the pair embedding, node rows = rows of the chemical embedding table (+ the charge-doping correction)"""
        z = data.z.contiguous()
        delta = rep.get("_charge_delta")                        # apply_charge_doping: node_attrs = one_hot(z) + delta
        g_emb = self.pair_embedding.backward(z, geo, g_f, chunk=chunk, delta=delta)
        g_delta = g_emb.pop("_g_delta", None)
        extra = rep.pop("_g_delta_extra", None)
        if extra is not None:
            g_delta = extra if g_delta is None else g_delta + extra.to(g_delta.dtype)
        grads.update({"pair_embedding." + k: v for k, v in g_emb.items()})
        T, lay = self.num_types, self.layout
        gtab = ops.scatter_rows(z, g_node, T)                   # fixed summation order (no float atomics)
        if delta is not None:                                   # node rows = (one_hot(z) + delta) @ table
            gtab = gtab + delta.t() @ g_node
            g_delta = g_delta + g_node @ self._chem.t()
            # the charge MLP (8 -> 8 -> num_types, torch tensor ops in the forward too): its parameters through autograd on those few ops
            with torch.enable_grad():
                d = self.atomic_embedding.delta(gget(data, "doping_charge"), gget(data, "batch"), z.shape[0], z.device)
                names, params = zip(*self.atomic_embedding.named_parameters())
                for k, gp in zip(names, torch.autograd.grad(d, params, grad_outputs=g_delta.to(d.dtype), allow_unused=True)):
                    grads["atomic_embedding." + k] = gp if gp is not None else torch.zeros_like(dict(self.atomic_embedding.named_parameters())[k])
        gw = [gtab[:, lay.off[k]:lay.off[k] + mk].reshape(-1) / math.sqrt(T) for k, (mk, lk, pk) in enumerate(self.irreps_node_features) if (lk, pk) == (0, 1)]
        grads["chemical_embedding.linear.weight"] = torch.cat(gw)


class HamGNNConvE3(_BackboneBase):
    def __init__(self, config):
        super().__init__()
        self.use_corr_prod = bool(_cfg_get(_cfg_get(config, "HamGNN_pre", config), "use_corr_prod", False))
        g = self._init_common(config)
        D, sh, R, mlp = self.irreps_node_features, self.irreps_edge_sh, self.num_radial, self.radial_MLP
        self.convolutions = nn.ModuleList()
        self.pair_interactions = nn.ModuleList()
        if self.use_corr_prod:                                  # hamgnn_conv.py:193-218
            self.corr_products = nn.ModuleList([hnn.CorrProductBlock(D, int(g("num_hidden_features")), int(g("correlation")), self.num_types, True)
                                                for _ in range(self.num_layers)])
        for i in range(self.num_layers):
            self.convolutions.append(hnn.ConvBlockE3(D, sh, R, mlp, self.lite_mode))
            skip = (i > 0) if self.legacy_edge_update else True
            self.pair_interactions.append(hnn.PairInteractionBlock(D, sh, R, mlp, skip, self.legacy_edge_update, self.lite_mode))
        self._mark_structural_zeros()

    def _mark_structural_zeros(self):
        """tell the message blocks of the leading layers which of their inputs are structurally zero (r5: the reference multiplies those zeros through all
        255 paths of both tensor products, message_passing.py:216-229; here the planner drops the super-paths that read them -- 74 % of the first
        ConvBlock launch and 20 % of the first PairInteractionBlock launch for the shipped irreps).  The node rows are full after the first ConvBlock,
        the edge rows after the first PairInteractionBlock that updates them (a legacy layer-0 block does not: interaction_blocks.py:154-160)."""
        zero_node, zero_edge = self._structural_zero_sets()
        for conv, pair in zip(self.convolutions, self.pair_interactions):
            conv.conv_tp.set_structural_zeros(node=zero_node, edge=zero_edge)
            zero_node = ()
            if pair.use_skip_connections or not pair.legacy_edge_update:
                pair.conv_tp.set_structural_zeros(node=(), edge=zero_edge)
                zero_edge = ()

    # ------------------------------------------------------------------------------------------------------------
    def compile(self, device):
        """(Re)pack all weights into MFMA fragment order and upload.  Call again after changing parameters."""
        dev = torch.device(device)
        for c, p in zip(self.convolutions, self.pair_interactions):
            c.compile(dev)
            p.compile(dev)
        if self.use_corr_prod:
            for c in self.corr_products:
                c.compile(dev)
        self._compile_common(dev)
        return self

    def refresh_weights(self):
        """after an optimiser step (hamgnn_amd.training): bring the uploaded programs up to date WITHOUT the host planner for the
        message blocks (device-side repack, hamgnn_amd/repack.py); the small Linear / embedding tables are rebuilt on the host (< 1 ms each)"""
        dev = self._compiled_for
        if dev is None:
            return
        if self.lite_mode:
            self._compiled_for = None                          # full recompile on the next forward
            return
        if self.use_corr_prod:
            for c in self.corr_products:
                c.compile(dev)                                 # four Linear tables + the concatenated element weights (host, ~ms)
        for conv, pair in zip(self.convolutions, self.pair_interactions):
            conv.residual.refresh(dev)                         # (its two Linears + the cached fused chain; the gate tables are structural)
            conv.skip_linear.compile(dev)
            if not conv.conv_tp.refresh():
                conv.conv_tp.compile(dev, unrotate=True)
            pair.refresh(dev)
        self._compile_common(dev)

    def forward(self, data, save_for_backward: bool = False):
        """save_for_backward: keep the layer inputs (node rows, edge rows, aggregated messages) on the result (`_tape`) for `backward`"""
        z, topo, geo, node, f = self._embed(data)
        N = z.shape[0]
        if os.environ.get("HG_CHECK_STRUCT_ZEROS") == "1":     # (tests) the blocks the first layer's programs treat as structural zeros ARE zero
            zn, ze = self._structural_zero_sets()
            for rows, idx in ((node, zn), (f, ze)):
                for i in idx:
                    o, w = self.layout.off[i], (2 * self.layout.irreps[i][1] + 1) * self.layout.mulp[i]
                    assert float(rows[:, o:o + w].abs().max()) == 0.0, ("structural zero violated", i)
        rowptr, perm = topo.receiver_csr()
        tape = [] if save_for_backward else None
        # the last PairInteractionBlock may leave out the irreps its declared consumer never reads (declare_consumer)
        last = self.pair_interactions[-1]
        has_dead, skip_dead = self._dead_plan(tape)
        for li, (conv, pair) in enumerate(zip(self.convolutions, self.pair_interactions)):
            # ---- ConvBlockE3.forward (convolution.py:116-160)
            row_shard = tape is None and parallel.node_shard_enabled(data)      # HG_NODE_SHARD=1: the node-level chain on this rank's block of rows only
            infer = tape is None                                # (inference: ResidualBlock as one row program + the skip Linear with the add in its epilogue, the two
            skip = None if (row_shard or infer) else conv.skip_linear(node)     #  linear_up Linears as one launch -- small crystals are launch-bound; training keeps the separate stages)
            if conv.conv_tp.can_reduce(geo.E):
                # convolution.py:147-149 fused into the edge kernel: receiver-major tiles, the runs of equal receivers summed in the epilogue
                # (about E / 13 rows instead of the [E, Dp] message tensor), then a segmented sum over each atom's contiguous rows
                eperm, run_id, R, prow, ident = topo.receiver_major()
                part = conv.conv_tp.run_nodes(node, node, f, geo, self._rot_tab, reduce=(eperm, run_id, R), structural_zeros=True)
                agg = ops.segment_sum(part, prow, ident, N)
            else:
                msg = conv.conv_tp.run_nodes(node, node, f, geo, self._rot_tab, structural_zeros=True)      # global frame (un-rotated in the epilogue)
                agg = ops.segment_sum(msg, rowptr, perm, N)
            if row_shard:
                # reduce-scatter of the partial aggregates, skip Linear / ResidualBlock / CorrProductBlock on N / world rows, all-gather of the new rows
                r0, r1, _ = parallel.node_rows(data, N)
                agg_r = parallel.reduce_scatter_nodes(agg, data)
                part = conv.residual(agg_r, skip=(conv.skip_linear, node[r0:r1].contiguous()))
                if self.use_corr_prod:
                    part = self.corr_products[li](part, z[r0:r1].contiguous(), None if self._last_delta is None else self._last_delta[r0:r1].contiguous())
                node = parallel.allgather_nodes(part, data, N)
                f_in, f = f, self._run_pair(pair, node, f, geo, reduced=(pair is not last) or skip_dead or not has_dead, merged_up=True)
                continue
            parallel.allreduce_nodes(agg, data)                                      # edge-sharded runs: RCCL sum over ranks
            if tape is not None:
                tape.append(dict(node_in=node, f_in=f, agg=agg))
            node = conv.residual(agg, skip=(conv.skip_linear, node)) if infer else conv.residual(agg, extra=skip)
            if self.use_corr_prod:                              # CorrProductBlock.forward (interaction_blocks.py:234-260; hamgnn_conv.py:274-275)
                if tape is not None:
                    tape[-1]["node_res"] = node                 # the ResidualBlock's output = the CorrProductBlock's input
                node = self.corr_products[li](node, z, self._last_delta)
            if tape is not None:
                tape[-1]["node_out"] = node
            f_in, f = f, self._run_pair(pair, node, f, geo, reduced=(pair is not last) or skip_dead or not has_dead, merged_up=infer)
        rep = self._representation(node, f, geo, (lambda: self._run_pair(last, node, f_in, geo, reduced=False)) if skip_dead else None)
        if tape is not None:
            rep["_tape"] = tape
            if self._last_delta is not None:
                rep["_charge_delta"] = self._last_delta
        return rep

    # ------------------------------------------------------------------------------------------------------------ backward
    def backward(self, data, rep, g_node, g_edge_rot, chunk: int = 65536):
        """Gradients of EVERY backbone parameter for the gradients of the representation the forward returned: g_node [N, Dp] (planar node
        rows) and g_edge_rot [E, Dp] (planar edge rows in the edge frame) -- what HamGNNPlusPlusOut.backward hands back.  `rep` must come
        from forward(data, save_for_backward=True).  Chains the block-level backwards (all on the HIP kernels + library GEMMs):
        per layer, last to first:  PairInteractionBlock (message block data + weight gradients, sender / receiver segment sums, the two
        linear_up adjoints, the fused skip o3.Linear)  ->  ResidualBlock  ->  skip o3.Linear  ->  ConvBlockE3's message block with the
        receiver scatter's adjoint (a gather) fused into its staging;  then the pair embedding and the chemical embedding table.
        Returns {reference parameter name: gradient in the reference's layout}.  Non-lite, no CorrProduct, no charge doping, one rank."""
        if parallel.is_sharded(data) and self.apply_charge_doping:
            raise NotImplementedError("backbone backward of an edge-sharded graph with charge doping")
        tape = rep["_tape"]
        geo = rep["_geometry"]
        z = data.z.contiguous()
        topo = get_topology(data)
        N = z.shape[0]
        rp_r, pm_r = topo.receiver_csr()
        rp_s, pm_s = topo.sender_csr()
        rot = self._rot_tab
        grads = {}
        put = lambda prefix, d: grads.update({prefix + k: v for k, v in d.items()})
        g_node, g_f = g_node.contiguous(), g_edge_rot.contiguous()
        for li in reversed(range(self.num_layers)):
            conv, pair, t = self.convolutions[li], self.pair_interactions[li], tape[li]
            node_in, f_in, agg, node_out = t["node_in"], t["f_in"], t["agg"], t["node_out"]
            g_node, g_f = self._backward_pair(li, pair, node_out, f_in, geo, topo, g_node, g_f, grads, chunk, data)
            # ---- ConvBlockE3 (convolution.py:116-160): node_out = residual(agg) + skip(node_in), agg = scatter_dst MP(node_in[src], node_in[dst], f_in)
            if self.use_corr_prod:                              # CorrProductBlock between the ConvBlock's residual and the pair block
                g_node, g_cp = self.corr_products[li].backward(t["node_res"], z, g_node, delta=rep.get("_charge_delta"))
                self._add_g_delta(rep, g_cp.pop("_g_delta", None))
                put(f"corr_products.{li}.", g_cp)
            pre = f"convolutions.{li}."
            g_agg, g_res = conv.residual.backward(agg, g_node, extra_given=True)
            put(pre + "residual.", g_res)
            grads[pre + "skip_linear.weight"] = conv.skip_linear.weight_grad(node_in, g_node)
            g_node_in = conv.skip_linear.backward_data(g_node)
            gs, gd, ge, g_tp = conv.conv_tp.backward(node_in, node_in, f_in, geo, rot, g_agg, out_is_global=True, gather=geo.dst, chunk=chunk, structural_zeros=True)
            put(pre + "conv_tp.", g_tp)
            part = ops.segment_sum(gs, rp_s, pm_s, N) + ops.segment_sum(gd, rp_r, pm_r, N)
            parallel.allreduce_nodes(part, data)               # edge-sharded: this rank's edges only -> all edges
            g_node = g_node_in + part
            g_f = g_f + ge
        self._backward_embeddings(data, rep, geo, g_node, g_f, grads, chunk)
        return grads

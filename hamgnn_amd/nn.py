"""Host-side mirror of the reference's hot-path modules (hamgnn/nn/*.py): same attribute / parameter names and flat
e3nn weight layouts (so reference state_dicts load unchanged), but `forward` drives the hand-written HIP kernels through
the C ABI.  Weights are (re)packed into MFMA fragment order by `compile()`.  The `backward*` methods are the block-level pieces of the
training path (chained by HamGNNConvE3.backward / HamGNNPlusPlusOut.backward / hamgnn_amd.training).

Reference classes mirrored here: o3.Linear / o3.TensorProduct / FullyConnectedNet parameter holders [e3nn 0.5.0];
LinearScaleWithWeights (tensor_products.py:25-47), TensorProductWithMemoryOptimizationWithWeight (:51-189),
MessagePackBlock (message_passing.py:26-231), ResidualBlock (interaction_blocks.py:264-358), ConvBlockE3
(convolution.py:22-160), PairInteractionBlock (interaction_blocks.py:30-164), PairInteractionEmbeddingBlock
(embeddings.py:215-337), HamLayer (models/hamgnn_output.py:38-58)."""
from __future__ import annotations

import os

import math
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import ops
from . import plan as P
from .so3 import Irreps


def _np_sd(module: nn.Module) -> Dict[str, np.ndarray]:
    return {k: v.detach().cpu().double().numpy() for k, v in module.state_dict().items()}


def o3_linear_weight_grad(irreps_in, irreps_out, x_planar: torch.Tensor, gy_planar: torch.Tensor) -> torch.Tensor:
    """d sum(y * gy) / d weight of an o3.Linear in e3nn's flat layout (paths (i_in, i_out), each [mul_in, mul_out], 1 / sqrt(fan_in)
    normalisation): per path one GEMM over (rows x components) on the planar blocks (rocBLAS / hipBLASLt: a library GEMM)."""
    irreps_in, irreps_out = Irreps(irreps_in), Irreps(irreps_out)
    if x_planar.is_cuda and os.environ.get("HG_LINEAR_WGRAD", "1") != "0":
        # all paths in one launch of csrc/linear_wgrad.hip (r4; the per-path library GEMMs on strided copies below were ~290 GEMMs + ~570 copy
        # launches of a training step): same sums, added in a fixed order
        xc = x_planar if x_planar.stride(1) == 1 else x_planar.contiguous()
        gc = gy_planar if gy_planar.stride(1) == 1 else gy_planar.contiguous()
        return ops.linear_wgrad(irreps_in, irreps_out, xc, gc)
    li, lo = P.PlanarLayout(irreps_in), P.PlanarLayout(irreps_out)
    paths = [(i, k) for i, (_, l1, p1) in enumerate(irreps_in) for k, (_, l2, p2) in enumerate(irreps_out) if (l1, p1) == (l2, p2)]
    fan = {}
    for i, k in paths:
        fan[k] = fan.get(k, 0) + irreps_in[i][0]
    rows = x_planar.shape[0]
    out = []
    for i, k in paths:
        mi, l, _ = irreps_in[i]
        mk = irreps_out[k][0]
        n = 2 * l + 1
        X = x_planar[:, li.off[i]:li.off[i] + n * li.mulp[i]].reshape(rows * n, li.mulp[i])[:, :mi]
        G = gy_planar[:, lo.off[k]:lo.off[k] + n * lo.mulp[k]].reshape(rows * n, lo.mulp[k])[:, :mk]
        out.append(((X.t() @ G) / math.sqrt(fan[k])).reshape(-1))
    return torch.cat(out) if out else x_planar.new_zeros(0)


_LINEAR_PACKERS: Dict[tuple, object] = {}


class E3Linear(nn.Module):
    """o3.Linear parameter holder: flat weight, paths ordered (i_in, i_out), each (mul_in, mul_out) row-major."""

    def __init__(self, irreps_in, irreps_out):
        super().__init__()
        self.irreps_in, self.irreps_out = Irreps(irreps_in), Irreps(irreps_out)
        n = sum(mi * mo for (mi, li, pi) in self.irreps_in for (mo, lo, po) in self.irreps_out if (li, pi) == (lo, po))
        self.weight = nn.Parameter(torch.randn(n))
        self._dp = None

    _stale = False                                             # the weights moved since the tables were packed (training._invalidate): refresh on next use

    def _packer(self, adjoint: bool):
        """blob = const + coef * weight[idx] of the (adjoint) streaming-Linear tables, discovered once per (irreps pair, direction) by probing the
        host builder (hamgnn_amd/repack.py) and shared by all Linears of that shape: after an optimiser step the tables are refreshed by one
        gather on the device -- no host planner, no device -> host copy of the weights (a synchronisation point per Linear and step before r3)"""
        from collections import OrderedDict
        from . import repack as RP
        key = (str(self.irreps_in), str(self.irreps_out), bool(adjoint))
        if key not in _LINEAR_PACKERS:
            build = P.build_linear_adjoint_tables if adjoint else P.build_linear_tables
            _LINEAR_PACKERS[key] = RP.AffinePack(lambda d: build(d["w"], self.irreps_in, self.irreps_out, keep_zero_blocks=True).weights,
                                                 OrderedDict(w=int(self.weight.numel())))
        return _LINEAR_PACKERS[key]

    def compile(self, device):
        self._stale = False
        stream = os.environ.get("HG_LINEAR_KERNEL", "stream") != "seg"
        if stream and isinstance(self._dp, ops.DeviceLinear) and self._dp.weights.device == torch.device(device) and getattr(self._dp, "refreshable", False):
            w = {"w": self.weight.detach()}                    # already uploaded: refresh the weight blobs in place, on the device
            self._dp.weights.copy_(self._packer(False).apply(w))
            if getattr(self, "_dp_adj", None) is not None and getattr(self._dp_adj, "refreshable", False):
                self._dp_adj.weights.copy_(self._packer(True).apply(w))
            else:
                self._dp_adj = None
            return self
        W = self.weight.detach().cpu().double().numpy()
        self._dp_adj = None                                    # adjoint tables are packed from the same weights: rebuilt on demand
        if not stream:                                         # HG_LINEAR_KERNEL=seg: the Linear as a program of the segment-stationary kernel
            self._dp = ops.DeviceProgram(P.build_linear_program(W, self.irreps_in, self.irreps_out), device)
        else:                                                  # default: the streaming block-Linear kernel (csrc/linear.hip)
            self._dp = ops.DeviceLinear(P.build_linear_tables(W, self.irreps_in, self.irreps_out, keep_zero_blocks=True), device)
            self._dp.refreshable = True
        return self

    def forward(self, x_planar: torch.Tensor, res=()) -> torch.Tensor:
        """res: residual rows (output layout) added in the kernel epilogue"""
        if self._dp is None or self._stale:
            self.compile(x_planar.device)
        if isinstance(self._dp, ops.DeviceLinear):
            return ops.linear_planar(self._dp, x_planar, res=res)
        return ops.tp_fused(self._dp, [x_planar], x_planar.shape[0], res=res)


    # ---- backward: data gradient on the same streaming kernel with transposed blocks; the weight gradient of a Linear is
    #      a plain reduction over the rows (one library GEMM per path)
    def backward_data(self, gy_planar: torch.Tensor) -> torch.Tensor:
        dev = gy_planar.device
        adj = getattr(self, "_dp_adj", None)
        if adj is not None and self._stale and getattr(adj, "refreshable", False) and self._dp is None:
            adj.weights.copy_(self._packer(True).apply({"w": self.weight.detach()}))    # (a Linear whose forward is fused elsewhere: only these tables exist)
            self._stale = False
        elif self._stale:
            self.compile(dev)
            adj = getattr(self, "_dp_adj", None)
        if adj is None:
            W = self.weight.detach().cpu().double().numpy()
            adj = self._dp_adj = ops.DeviceLinear(P.build_linear_adjoint_tables(W, self.irreps_in, self.irreps_out, keep_zero_blocks=True), dev)
            adj.refreshable = True
        return ops.linear_planar(adj, gy_planar, tag="linear_adjoint")

    def weight_grad(self, x_planar: torch.Tensor, gy_planar: torch.Tensor) -> torch.Tensor:
        return o3_linear_weight_grad(self.irreps_in, self.irreps_out, x_planar, gy_planar)


class E3TensorProduct(nn.Module):
    """o3.TensorProduct(uvw, internal shared weights) parameter holder; instructions by the reference rule."""

    def __init__(self, irreps_in1, irreps_sh, irreps_target):
        super().__init__()
        i1, i2, io = Irreps(irreps_in1), Irreps(irreps_sh), Irreps(irreps_target)
        self.ins = P.tp_instructions(i1, i2, io)
        self.weight_numel = sum(i1[i][0] * i2[j][0] * io[k][0] for i, j, k, _ in self.ins)
        self.mid_num_irreps = sum(io[k][0] for _, _, k, _ in self.ins)
        self.mid_fan = {}
        for _, _, k, _ in self.ins:
            self.mid_fan[k] = self.mid_fan.get(k, 0) + io[k][0]
        self.weight = nn.Parameter(torch.randn(self.weight_numel))


class _FCLayer(nn.Module):
    def __init__(self, h_in, h_out):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(h_in, h_out))


class FullyConnectedNet(nn.Module):
    def __init__(self, hs):
        super().__init__()
        self.hs = list(hs)
        for i, (a, b) in enumerate(zip(hs, hs[1:])):
            setattr(self, f"layer{i}", _FCLayer(a, b))

    def hidden_layers(self, device):
        n = len(self.hs) - 1
        out = []
        for i in range(n - 1):
            W = getattr(self, f"layer{i}").weight.detach().double()
            W = W / math.sqrt(W.shape[0])
            if i == n - 2 and W.shape[1] % 16:                 # pad the last hidden width to the kernel's K granule (silu(0)=0)
                W = torch.nn.functional.pad(W, (0, 16 - W.shape[1] % 16))
            out.append(W.float().contiguous().to(device))
        return out


class _LinOutHolder(nn.Module):
    def __init__(self, tp: E3TensorProduct, irreps_out: Irreps):
        super().__init__()
        n = sum(tp.mid_fan[k] * irreps_out[k][0] for k in tp.mid_fan)
        self.weight = nn.Parameter(torch.randn(n))


class LinearScaleWithWeights(nn.Module):
    def __init__(self, tp: E3TensorProduct, irreps_out: Irreps):
        super().__init__()
        self.weight_numel = tp.mid_num_irreps
        self.linear_out = _LinOutHolder(tp, irreps_out)


class _MidLinear(nn.Module):
    """o3.Linear(mid.simplify() -> irreps_out) of the lite-mode branches: weight holder (fan_k = sum of the path input muls)."""

    def __init__(self, irreps_in1: Irreps, irreps_sh: Irreps, irreps_out: Irreps):
        super().__init__()
        ins = P.tp_instructions(irreps_in1, irreps_sh, irreps_out)
        fan = {}
        for i, _, k, _ in ins:
            fan[k] = fan.get(k, 0) + irreps_in1[i][0]
        self.weight = nn.Parameter(torch.randn(sum(f * irreps_out[k][0] for k, f in fan.items())))


class _CombineMessages(nn.Module):
    def __init__(self, irreps_out: Irreps):
        super().__init__()
        self.weight_numel = irreps_out.simplify().num_irreps
        self.linear_out = E3Linear(irreps_out.simplify(), irreps_out)


# "auto": input-stationary schedule (csrc/tp_is.hip) when the tiles of all output segments fit the LDS, else segment-stationary
MP_KERNEL_DEFAULT = "auto"


def _wgrad_runner(wg, dpA, dpB):
    """backward_mp's run_program on the device: the two materialisation programs on the segment-stationary fused kernel"""
    dps = {id(wg.progA): dpA, id(wg.progB): dpB}
    return lambda prog, srcs, hn, he: ops.tp_fused(dps[id(prog)], [t.contiguous() for t in srcs], srcs[0].shape[0], hn, he, None, tag="wgrad_rows")


class MessagePackBlock(nn.Module):
    def __init__(self, irreps_node_feats, irreps_edge_feats, irreps_local_env_edge, irreps_out, num_radial, radial_MLP=(64, 64),
                 lite_mode=False):
        super().__init__()
        self.lite_mode = lite_mode
        self.irreps_node, self.irreps_edge = Irreps(irreps_node_feats), Irreps(irreps_edge_feats)
        self.irreps_sh, self.irreps_out = Irreps(irreps_local_env_edge), Irreps(irreps_out)
        comb = Irreps([(max(1, 2 * m), l, p) for m, l, p in self.irreps_node])
        if lite_mode:                                          # message_passing.py:99-111, 123-125 (uvu products carry no weights)
            self.node_linear_scaler = _MidLinear(comb, self.irreps_sh, self.irreps_out)
            self.edge_linear_scaler = _MidLinear(self.irreps_edge, self.irreps_sh, self.irreps_out)
            self.combine_messages = _CombineMessages(self.irreps_out)
            self.weight_generator_combine = FullyConnectedNet([num_radial] + list(radial_MLP) + [self.combine_messages.weight_numel])
        else:
            self.node_tensor_product = E3TensorProduct(comb, self.irreps_sh, self.irreps_out)
            self.edge_tensor_product = E3TensorProduct(self.irreps_edge, self.irreps_sh, self.irreps_out)
            self.node_linear_scaler = LinearScaleWithWeights(self.node_tensor_product, self.irreps_out)
            self.edge_linear_scaler = LinearScaleWithWeights(self.edge_tensor_product, self.irreps_out)
            self.node_weight_generator = FullyConnectedNet([num_radial] + list(radial_MLP) + [self.node_linear_scaler.weight_numel])
            self.edge_weight_generator = FullyConnectedNet([num_radial] + list(radial_MLP) + [self.edge_linear_scaler.weight_numel])
            self.node_linear_out = E3Linear(self.irreps_out, self.irreps_out)
            self.edge_linear_out = E3Linear(self.irreps_out, self.irreps_out)
        self._dp = None

    def set_structural_zeros(self, node=(), edge=()):
        """irreps (indices into irreps_node / irreps_edge) whose rows are STRUCTURALLY zero where this block runs (the first layer: node rows out of an
        o3.Linear from 0e scalars, edge rows out of the pair embedding's 0e x Y^l product): the forward program drops the super-paths that read them
        (plan.build_message_pack_program).  Takes effect at the next compile(); the backward programs stay complete (they produce the zeros)."""
        self._zeros = (tuple(sorted(int(i) for i in node)), tuple(sorted(int(i) for i in edge)))
        return self

    def set_dead_outputs(self, out=()):
        """irreps (indices into irreps_out) of this block's result that NOBODY reads where it runs (the last PairInteractionBlock of a backbone whose only
        consumer is a read-out head that declared what it reads: HamGNNConvE3.declare_consumer).  The reduced program (`structural_zeros=True` launches) drops
        their super-paths and writes zeros there; the generic program stays complete.  Takes effect at the next compile()."""
        self._dead = tuple(sorted(int(k) for k in out))
        return self

    def _zero_kw(self):
        """what the REDUCED forward program of this block may assume (plan.build_message_pack_program): structurally zero input irreps, unread output irreps"""
        zn, ze = getattr(self, "_zeros", ((), ()))
        dead = getattr(self, "_dead", ())
        if self.lite_mode:
            return {}
        kw = {}
        if os.environ.get("HG_STRUCT_ZEROS", "1") != "0" and (zn or ze):
            kw.update(zero_node=zn, zero_edge=ze)
        if os.environ.get("HG_DEAD_OUT", "1") != "0" and dead:
            kw["dead_out"] = dead
        return kw

    def _zero_inputs_kw(self):
        """the structurally zero INPUT irreps alone (what the backward programs may assume: paths that read zeros have zero weight gradients, and nobody
        reads the gradient of a structurally zero input)"""
        return {k: v for k, v in self._zero_kw().items() if k in ("zero_node", "zero_edge")}

    def compile(self, device, unrotate: bool, skip_weight=None):
        sd = _np_sd(self)
        zkw = self._zero_kw()
        self._zkw_compiled = dict(zkw)                         # what the reduced program of THIS compile assumes (callers plan with it, not with the environment of the moment)
        self._dp_z = self._dp_z_plain = None                    # programs for rows with structurally zero irreps (set_structural_zeros), built next to the generic ones
        self._compile_args = (bool(unrotate), skip_weight is not None, None)      # (unrotate, fused skip Linear, merge groups)
        self._lite_bw = None
        self._packers = getattr(self, "_packers", None) or {}                     # structural: survive recompiles of the same block
        self._dp_adj = self._dp_adj_z = None                   # the data-gradient programs are packed from the same weights
        self._wgrad_prev, self._wgrad = (getattr(self, "_wgrad", None) or getattr(self, "_wgrad_prev", None)), None
        self._wgrad_fused = self._wgrad_fused_z = None         # (tables hold the weights: rebuilt on first use)
        if self.lite_mode:
            if skip_weight is not None:                        # PairInteractionBlock skip o3.Linear: must come AFTER the combine post-op
                raise NotImplementedError
            self._hn = self.weight_generator_combine.hidden_layers(device)
            self._he = None
            self._plain_args = None
            if os.environ.get("HG_MP_KERNEL", MP_KERNEL_DEFAULT) != "seg":
                # input-stationary kernel (r3): the paths of every (input irrep, output irrep) pair folded into one item (IT_LINM), the
                # combine post-op as the last phase; such a program has no segment-stationary form
                try:
                    args = (sd, self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, unrotate)
                    self._dp = ops.DeviceProgram(P.build_message_pack_program_lite(*args, fold=os.environ.get("HG_LITE_FOLD", "1") != "0"), device, schedule="is")
                    return self
                except NotImplementedError:
                    pass
            prog = P.build_message_pack_program_lite(sd, self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, unrotate)
            self._dp = ops.DeviceProgram(prog, device, schedule="seg")
            return self
        else:
            self._hn = self.node_weight_generator.hidden_layers(device)
            self._he = self.edge_weight_generator.hidden_layers(device)
            sched = os.environ.get("HG_MP_KERNEL", MP_KERNEL_DEFAULT)
            if sched != "seg" and os.environ.get("HG_MP_MERGE", "1") != "0":
                # input-stationary kernel with the small output irreps sharing MFMA row tiles (plan.choose_merge_groups): -5 % MFMAs, -17 %
                # items for set-A.  Such a program has no segment-stationary form: if it does not fit, fall back to the plain program.
                groups = P.choose_merge_groups(self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, self._hn[-1].shape[1])
                if groups:
                    try:
                        prog = P.build_message_pack_program(sd, self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, unrotate,
                                                            skip_weight, merge_groups=groups)
                        self._dp = ops.DeviceProgram(prog, device, schedule="is")
                        self._compile_args = (bool(unrotate), skip_weight is not None, groups)
                        if zkw:                                # the same block for rows whose marked irreps are structurally zero (first layer of a backbone) /
                            gz = groups                        # whose marked output irreps nobody reads (last PairInteractionBlock): those leave the row-tile groups
                            if zkw.get("dead_out"):
                                gz = P.choose_merge_groups(self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, self._hn[-1].shape[1], dead_out=zkw["dead_out"])
                            self._groups_z = gz
                            self._dp_z = ops.DeviceProgram(P.build_message_pack_program(sd, self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, unrotate,
                                                                                       skip_weight, merge_groups=gz, **zkw), device, schedule="is")
                        # launches with fewer 16-edge tiles than workgroup slots run the PLAIN program split over one workgroup per output
                        # segment: merging trades parts (9 instead of 13 for set-A) for MFMAs, the wrong trade when latency is all there is
                        # (Si 2-atom cell: 0.113 -> 0.110 ms per launch); built on first use
                        self._plain_args = (sd, unrotate, skip_weight, device)
                        self._dp_plain = None
                        return self
                    except NotImplementedError:
                        pass
            prog = P.build_message_pack_program(sd, self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, unrotate, skip_weight)
            if zkw:
                self._dp_z = ops.DeviceProgram(P.build_message_pack_program(sd, self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, unrotate, skip_weight, **zkw),
                                               device, schedule=os.environ.get("HG_MP_KERNEL", MP_KERNEL_DEFAULT))
        self._plain_args = None
        # HG_MP_KERNEL = seg | is | auto: which schedule of the fused MessagePackBlock program runs (default: see DESIGN.md section 5)
        self._dp = ops.DeviceProgram(prog, device, schedule=os.environ.get("HG_MP_KERNEL", MP_KERNEL_DEFAULT))
        return self

    # ---- training: repack the weights of the uploaded programs on the device after an optimiser step (hamgnn_amd/repack.py)
    def refresh(self, skip=None) -> bool:
        """skip: the flat weight of the fused skip o3.Linear (device tensor) if the block was compiled with one.  Returns False when
        there is nothing to refresh in place (never compiled, lite_mode): the caller compiles instead."""
        from . import repack as RP
        if self.lite_mode or self._dp is None or getattr(self, "_compile_args", None) is None:
            return False
        unrotate, has_skip, groups = self._compile_args
        if has_skip != (skip is not None):
            return False
        params = {k: v.detach() for k, v in self.state_dict().items()}
        shapes = {k: tuple(v.shape) for k, v in params.items()}
        last = {n: sorted(k for k in params if k.startswith(f"{n}_weight_generator.layer") and k.endswith(".weight"))[-1] for n in ("node", "edge")}
        lays = RP.mp_branch_layouts(self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out)
        args = (self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out)
        nskip = int(skip.numel()) if skip is not None else 0
        lp = None
        dev0 = self._dp.weights.device
        if ops.use_block_gemm(self._dp.weights):               # the 13 L' products of a branch in ONE launch (csrc/block_gemm.hip) instead of 13 GEMMs
            if getattr(self, "_bg_lp", None) is None or self._bg_lp[0] != dev0:
                self._bg_lp = (dev0, {name: (ops.BlockGemm(RP.lp_block_units(lay)[0], dev0), RP.lp_block_units(lay)[1]) for name, lay in lays})
            lp = {}
            for name, (bg, total) in self._bg_lp[1].items():
                ls, lo = params[f"{name}_linear_scaler.linear_out.weight"].float().reshape(-1).contiguous(), params[f"{name}_linear_out.weight"].float().reshape(-1).contiguous()
                lp[name] = ops.block_gemm(bg, ls, lo, torch.empty(total, device=dev0, dtype=torch.float64))
        src = RP.mp_sources(lambda k: params[k].double().reshape(-1), last, lays, None if skip is None else skip.detach().double().reshape(-1), lib=torch, lp=lp)

        def packer(tag, fn, ns):
            if tag not in self._packers:
                sizes = RP.mp_source_sizes(shapes, last, ns)
                self._packers[tag] = RP.AffinePack(lambda d: fn(RP.mp_probe_state_dict(d, shapes, last, lays, self.irreps_out), d.get("skip")), sizes)
            return self._packers[tag]

        def update(dp, tag, fn, ns):
            blob = packer(tag, fn, ns).apply({k: v for k, v in src.items() if ns or k != "skip"})
            assert blob.numel() == dp.weights.numel(), (tag, blob.numel(), dp.weights.numel())
            dp.weights.copy_(blob)
            dp.weights_changed()

        zkw = self._zero_kw()
        ztag = (zkw.get("zero_node", ()), zkw.get("zero_edge", ()), zkw.get("dead_out", ()))
        fwd = lambda g_, z_=None: (lambda d, sk: P.build_message_pack_program(d, *args, unrotate, sk, **({"merge_groups": g_} if g_ else {}), **(z_ or {})).weights)
        update(self._dp, ("fwd", unrotate, has_skip, bool(groups)), fwd(groups), nskip)
        if getattr(self, "_dp_plain", None) is not None and self._dp_plain is not self._dp:
            update(self._dp_plain, ("fwd", unrotate, has_skip, False), fwd(None), nskip)
        if getattr(self, "_dp_z", None) is not None:
            gz = getattr(self, "_groups_z", groups) if groups else groups
            update(self._dp_z, ("fwd", unrotate, has_skip, bool(gz), ztag), fwd(gz, zkw), nskip)
        if getattr(self, "_dp_z_plain", None) is not None and self._dp_z_plain is not self._dp_z:
            update(self._dp_z_plain, ("fwd", unrotate, has_skip, False, ztag), fwd(None, zkw), nskip)
        if getattr(self, "_dp_adj", None) is not None:
            update(self._dp_adj, ("adj",), lambda d, sk: P.build_message_pack_adjoint_program(d, *args).weights, 0)
        zin = self._zero_inputs_kw()
        if getattr(self, "_dp_adj_z", None) is not None:
            update(self._dp_adj_z, ("adj", ztag[:2]), lambda d, sk: P.build_message_pack_adjoint_program(d, *args, **zin).weights, 0)
        if getattr(self, "_wgrad", None) is not None:
            wg, dpA, dpB = self._wgrad
            if dpA is not None:
                update(dpA, ("wgA",), lambda d, sk: P.build_message_pack_wgrad_programs(d, *args)[0].weights, 0)
                update(dpB, ("wgB",), lambda d, sk: P.build_message_pack_wgrad_programs(d, *args)[1].weights, 0)
            wg.params_dev = params                             # the radial MLP / Ls / Lo values that the reductions read: straight from the device
            dwf = getattr(self, "_wgrad_fused", None)
            if dwf:
                irr = (self.irreps_node, self.irreps_edge)
                fused = lambda d, sk: P.build_tp_wgrad_fused(P.message_pack_wgrad_branches(d, *irr), self.irreps_sh, self.irreps_out, dwf.wf.hidden).weights
                dwf.weights.copy_(packer(("wgF",), fused, 0).apply({k: v for k, v in src.items() if k != "skip"}))
            dwz = getattr(self, "_wgrad_fused_z", None)
            if dwz:
                irr = (self.irreps_node, self.irreps_edge)
                zi = {"node": zin.get("zero_node", ()), "edge": zin.get("zero_edge", ())}
                fused_z = lambda d, sk: P.build_tp_wgrad_fused(P.message_pack_wgrad_branches(d, *irr), self.irreps_sh, self.irreps_out, dwz.wf.hidden, zero_inputs=zi).weights
                dwz.weights.copy_(packer(("wgF", ztag[:2]), fused_z, 0).apply({k: v for k, v in src.items() if k != "skip"}))
        dev = self._dp.weights.device
        self._hn = self.node_weight_generator.hidden_layers(dev)
        self._he = self.edge_weight_generator.hidden_layers(dev)
        return True

    # ---- backward: data gradient as an adjoint program, weight gradients through backward_mp
    def compile_adjoint(self, device, structural_zeros: bool = False):
        """upload the data-gradient program of this block (plan.build_message_pack_adjoint_program): same kernels, same weights.  structural_zeros: the
        variant that does not compute the gradient of the structurally zero input irreps (set_structural_zeros; nobody reads it)"""
        if self.lite_mode:
            raise NotImplementedError("data gradient of a lite_mode MessagePackBlock")
        zin = self._zero_inputs_kw() if structural_zeros else {}
        prog = P.build_message_pack_adjoint_program(_np_sd(self), self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, **zin)
        try:
            dp = ops.DeviceProgram(prog, device, schedule="is_parts" if os.environ.get("HG_MP_KERNEL", MP_KERNEL_DEFAULT) != "seg" else "seg")
        except NotImplementedError:                            # tiles / staging do not fit even split over workgroups: segment-stationary kernel
            dp = ops.DeviceProgram(prog, device, schedule="seg")
        setattr(self, "_dp_adj_z" if zin else "_dp_adj", dp)
        _, maps = P.message_pack_adjoint_layout(self.irreps_node, self.irreps_edge)
        self._adj_maps = tuple(torch.from_numpy(m).to(device) for m in maps)
        return self

    def backward_data(self, grad_out, geo: ops.Geometry, out_is_global: bool, gather=None, structural_zeros: bool = False):
        """grad_out [E, planar(irreps_out)]: gradient with respect to the rows this block's forward returned (global frame if the block
        was compiled with unrotate=True, else edge frame); or, with `gather` = an [E] index tensor, NODE rows whose gather is that
        per-edge gradient (the backward of the receiver scatter of a ConvBlockE3 is the gather grad_agg[receiver]; it is fused into the
        kernel's staging like the forward's node gathers).  Returns per-edge gradients (g_src_rows, g_dst_rows, g_edge_rows), planar:
        the first two in the GLOBAL frame, to be summed over the edges of each sender / receiver (ops.segment_sum over the sender /
        receiver CSR) for the gradient of the gathered node rows; the third in the edge frame, where the forward read the edge rows."""
        z = bool(structural_zeros and self._zero_inputs_kw())   # (the caller vouches as for run_nodes: the marked input irreps are zero, their gradient unread)
        slot = "_dp_adj_z" if z else "_dp_adj"
        if getattr(self, slot, None) is None:
            self.compile_adjoint(grad_out.device, structural_zeros=z)
        cst = float(P.ACT_CONSTS[P.ACT_SILU])
        hn = ops.radial_hidden_cached(geo, self._hn, cst)
        he = ops.radial_hidden_cached(geo, self._he, cst)
        dp = getattr(self, slot)
        if dp.sched is not None:
            g = ops.tp_fused(dp, [grad_out], geo.E, hn, he, geo, tag="message_pack_adjoint", gather=[gather], rot_mask=1 if out_is_global else 0)
        else:
            if out_is_global:
                src = ops.rotate_gather(grad_out, gather, geo, self._rot_tab_out(grad_out.device))
            else:
                src = grad_out if gather is None else grad_out[gather].contiguous()
            g = ops.tp_fused(dp, [src], geo.E, hn, he, geo, tag="message_pack_adjoint")
        ims, imd, ime = self._adj_maps
        return ops.from_planar(g, ims), ops.from_planar(g, imd), ops.from_planar(g, ime)      # column gathers (-1 = padding slot -> 0)

    def _dp_for(self, rows: int, structural_zeros: bool = False):
        """the program a launch of `rows` edges runs: the merged one, or -- split launches of small crystals -- the plain one; structural_zeros: the caller
        vouches that the irreps marked by set_structural_zeros are zero in the rows it passes (the backbone's first layer) AND that nobody reads the output
        irreps marked by set_dead_outputs (they come back as zeros): the reduced program"""
        z = structural_zeros and getattr(self, "_dp_z", None) is not None
        dp = self._dp_z if z else self._dp
        if getattr(self, "_plain_args", None) is None or dp.is_parts_for(rows) == 1:
            return dp
        slot = "_dp_z_plain" if z else "_dp_plain"
        if getattr(self, slot, None) is None:
            _, unrotate, skip_weight, device = self._plain_args
            sd = _np_sd(self)                                  # the CURRENT weights (the merged program may have been refreshed since compile())
            if skip_weight is not None and getattr(self, "_skip_source", None) is not None:
                skip_weight = self._skip_source[0].weight.detach().cpu().double().numpy()
            prog = P.build_message_pack_program(sd, self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out, unrotate, skip_weight, **(self._zero_kw() if z else {}))
            try:
                setattr(self, slot, ops.DeviceProgram(prog, device, schedule="is"))
            except NotImplementedError:
                setattr(self, slot, dp)
        return getattr(self, slot)

    def backward_weights(self, node_s, node_d, f_rot, geo: ops.Geometry, rot_tab, grad_out, out_is_global: bool, chunk: int = 65536, gather=None,
                         structural_zeros: bool = False):
        """gradients of every parameter of this block for the output gradient `grad_out` (frame and `gather` as in backward_data), first
        version (hamgnn_amd/backward_mp.py): two materialisation programs on the fused kernels + library GEMMs over the edges.
        node_s / node_d: planar NODE rows gathered by sender / receiver as in run_nodes; f_rot: planar edge rows (edge frame).
        Returns {reference parameter name: gradient in the reference's flat layout}."""
        from . import backward_mp as BM
        if self.lite_mode:
            raise NotImplementedError("weight gradients of a lite_mode MessagePackBlock")
        dev = grad_out.device
        if getattr(self, "_wgrad", None) is None:
            wg = BM.MessagePackWeightGrad(_np_sd(self), self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out)
            wg.adopt_constants(self._wgrad_prev[0] if getattr(self, "_wgrad_prev", None) else None)
            wg.params_dev = {k: v.detach() for k, v in self.state_dict().items()}
            self._wgrad = [wg, None, None]
        wg, dpA, dpB = self._wgrad
        xs, xd = ops.rotate_gather(node_s, geo.src, geo, rot_tab, x2=node_d, idx2=geo.dst)
        if out_is_global:
            g = ops.rotate_gather(grad_out, gather, geo, self._rot_tab_out(dev))
        else:
            g = grad_out if gather is None else grad_out[gather].contiguous()
        cst = float(P.ACT_CONSTS[P.ACT_SILU])
        dwf = self._wgrad_fused_for(wg, dev, structural_zeros)
        if dwf is not None:                                    # fused kernel (csrc/tp_wgrad.hip): nothing per edge is materialised but gs
            hidden = {"node": ops.radial_hidden_cached(geo, self._hn, cst), "edge": ops.radial_hidden_cached(geo, self._he, cst)}
            run = lambda srcs, g_, hn, he: ops.tp_wgrad(dwf, srcs, g_, hn, he)
            return BM.tp_weight_grads_fused(wg, dwf, run, [xs, xd, f_rot], g, geo.rbf, cst, hidden=hidden)     # dwf: the gather maps as device tensors
        if dpA is None:                                        # materialisation route: the two row programs on the segment-stationary kernel
            dpA, dpB = ops.DeviceProgram(wg.progA, dev, schedule="seg"), ops.DeviceProgram(wg.progB, dev, schedule="seg")
            self._wgrad[1:] = [dpA, dpB]
        return BM.block_weight_grads(wg, _wgrad_runner(wg, dpA, dpB), xs, xd, f_rot, g, geo.rbf, cst, chunk=chunk)

    def _wgrad_fused_for(self, wg, dev, structural_zeros: bool = False):
        """the fused weight-gradient tables of this block on the device, or None (HG_WGRAD=rows, or no kernel instantiation for these irreps:
        the materialisation route then).  structural_zeros: the tables without the row tiles of super-paths that read structurally zero input irreps
        (their gradients are exactly zero: plan.build_tp_wgrad_fused)"""
        if os.environ.get("HG_WGRAD", "fused") != "fused":
            return None
        zin = self._zero_inputs_kw() if structural_zeros else {}
        slot = "_wgrad_fused_z" if zin else "_wgrad_fused"
        cur = getattr(self, slot, None)
        if cur is None:
            try:
                zi = {"node": zin.get("zero_node", ()), "edge": zin.get("zero_edge", ())} if zin else None
                wf = P.build_tp_wgrad_fused(wg.branches, self.irreps_sh, self.irreps_out, wg.H, zero_inputs=zi)
                cur = ops.DeviceWgFused(wf, dev)
            except NotImplementedError:
                cur = False
            setattr(self, slot, cur)
        return cur or None

    def backward(self, node_s, node_d, f_rot, geo: ops.Geometry, rot_tab, grad_out, out_is_global: bool, gather=None, chunk: int = 65536,
                 structural_zeros: bool = False):
        """data AND weight gradients of the block in one call: (g_src_rows, g_dst_rows, g_edge_rows, {parameter name: gradient}); arguments
        as backward_data / backward_weights.  lite_mode blocks go through hamgnn_amd/backward_lite.py (nothing large to materialise).
        structural_zeros: as run_nodes -- the caller vouches that the input irreps marked by set_structural_zeros are zero in the rows it passes AND that
        it does not read their gradient (a backbone's first layer): weight gradients of the paths that read them are exactly zero and not computed."""
        if not self.lite_mode:
            grads = self.backward_weights(node_s, node_d, f_rot, geo, rot_tab, grad_out, out_is_global, chunk=chunk, gather=gather, structural_zeros=structural_zeros)
            return self.backward_data(grad_out, geo, out_is_global, gather=gather, structural_zeros=structural_zeros) + (grads,)
        from . import backward_lite as BL
        dev = grad_out.device
        if getattr(self, "_lite_bw", None) is None:
            lb = BL.LiteBackward(_np_sd(self), self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out)
            dp_t = ops.DeviceProgram(lb.prog_t, dev, schedule="seg")
            try:
                dp_a = ops.DeviceProgram(lb.prog_adj, dev, schedule="is_parts" if os.environ.get("HG_MP_KERNEL", MP_KERNEL_DEFAULT) != "seg" else "seg")
            except NotImplementedError:
                dp_a = ops.DeviceProgram(lb.prog_adj, dev, schedule="seg")
            _, maps = P.message_pack_adjoint_layout(self.irreps_node, self.irreps_edge)
            self._lite_bw = (lb, {id(lb.prog_t): dp_t, id(lb.prog_adj): dp_a}, ops.DeviceLinear(lb.lc_adj, dev),
                             tuple(torch.from_numpy(m).to(dev) for m in maps))
        lb, dps, dl, maps = self._lite_bw
        xs, xd = ops.rotate_gather(node_s, geo.src, geo, rot_tab, x2=node_d, idx2=geo.dst)
        if out_is_global:
            g = ops.rotate_gather(grad_out, gather, geo, self._rot_tab_out(dev))
        else:
            g = grad_out if gather is None else grad_out[gather].contiguous()
        hn = ops.radial_hidden_cached(geo, self._hn, float(P.ACT_CONSTS[P.ACT_SILU]))   # (no item of the two programs reads it; never hand the kernel a NULL row pointer)
        run_program = lambda prog, srcs: ops.tp_fused(dps[id(prog)], [t.contiguous() for t in srcs], geo.E, hn, None, geo, tag="lite_backward")
        run_linear = lambda tabs, x: ops.linear_planar(dl, x.contiguous(), tag="linear_adjoint")
        rows, grads = lb.run(run_program, run_linear, o3_linear_weight_grad, xs, xd, f_rot, g, geo.rbf, float(P.ACT_CONSTS[P.ACT_SILU]),
                             params={k: v.detach() for k, v in self.state_dict().items()})
        return ops.from_planar(rows, maps[0]), ops.from_planar(rows, maps[1]), ops.from_planar(rows, maps[2]), grads

    def _rot_tab_out(self, device):
        if getattr(self, "_rt_out", None) is None:
            self._rt_out = torch.from_numpy(P.rotate_table(P.PlanarLayout(self.irreps_out))).to(device)
        return self._rt_out

    def run(self, xs_rot, xd_rot, f_rot, geo: ops.Geometry):
        """xs_rot/xd_rot/f_rot: planar rows in the edge-aligned frame.  Returns planar [E, Dp] (global frame if unrotate)."""
        cst = float(P.ACT_CONSTS[P.ACT_SILU])
        hn = ops.radial_hidden_cached(geo, self._hn, cst)
        he = ops.radial_hidden_cached(geo, self._he, cst) if self._he is not None else None
        return ops.tp_fused(self._dp_for(geo.E), [xs_rot, xd_rot, f_rot], geo.E, hn, he, geo, tag="message_pack")

    def can_reduce(self, rows: int) -> bool:
        """the fused node scatter is a feature of the single-part input-stationary launch (large graphs)"""
        return self._dp.sched is not None and self._dp_for(rows).is_parts_for(rows) == 1 and os.environ.get("HG_FUSED_SCATTER", "1") != "0"

    def run_nodes(self, node_s, node_d, f_rot, geo: ops.Geometry, rot_tab, reduce=None, structural_zeros: bool = False):
        """node_s / node_d: planar NODE rows (global frame) whose sender / receiver gathers feed the block
        (convolution.py:138-141, interaction_blocks.py:141-145).  Input-stationary schedule: gathered and rotated inside the kernel;
        otherwise through hg_rotate_gather."""
        if self._dp.sched is None:
            xs, xd = ops.rotate_gather(node_s, geo.src, geo, rot_tab, x2=node_d, idx2=geo.dst)
            return self.run(xs, xd, f_rot, geo)
        cst = float(P.ACT_CONSTS[P.ACT_SILU])
        hn = ops.radial_hidden_cached(geo, self._hn, cst)
        he = ops.radial_hidden_cached(geo, self._he, cst) if self._he is not None else None
        return ops.tp_fused(self._dp_for(geo.E, structural_zeros), [node_s, node_d, f_rot], geo.E, hn, he, geo, tag="message_pack", gather=[geo.src, geo.dst, None],
                            rot_mask=0b011, reduce=reduce)


class ResidualBlock(nn.Module):
    def __init__(self, irreps_in, feature_irreps_hidden, resnet=True, nonlinearity_type="gate"):
        """interaction_blocks.py:262-358.  nonlinearity_type "gate" (e3nn Gate; every block of the backbone, the head's default) or "norm" (e3nn
        NormActivation with ShiftedSoftPlus as given, normalize, epsilon 1e-8: the head's HamLayers under `nonlinearity_type: norm`, :311-330)"""
        super().__init__()
        if nonlinearity_type not in ("gate", "norm"):
            raise AssertionError("Invalid nonlinearity_type. Choose either 'gate' or 'norm'.")      # (the reference asserts, :289-290)
        self.irreps_in = Irreps(irreps_in)
        self.nonlinearity_type = nonlinearity_type
        if nonlinearity_type == "norm":
            self.gate_in = self.gate_out = Irreps(feature_irreps_hidden)
            self._tab_np = P.norm_act_table(self.gate_in)
        else:
            self.gate_in, self.gate_out, self._tab_np = P.gate_tables(feature_irreps_hidden)
        self.linear1 = E3Linear(self.irreps_in, self.gate_in)
        self.linear2 = E3Linear(self.gate_out, self.irreps_in)
        self.resnet = resnet
        self._tab = None

    def compile(self, device):
        self.linear1.compile(device)
        self.linear2.compile(device)
        if self.nonlinearity_type == "norm":
            self._tab = torch.from_numpy(self._tab_np).contiguous().to(device)
        else:
            self._tab = tuple(torch.from_numpy(t).contiguous().to(device) for t in P.gate_tables_compact(self._tab_np))
        self._cst = torch.from_numpy(P.ACT_CONSTS).to(device)
        self._rowprog = None                                   # the fused inference chain is built on first use, from the weights of that moment
        self._rowprog_off = getattr(self, "_rowprog_off", False)   # set by training._invalidate: separate kernels while the weights move
        return self

    def refresh(self, device):
        """after an optimiser step: the two Linears' tables rebuilt on the host (the gate tables are structural) and the cached fused chain DROPPED -- it was
        built from the weights of its moment, so validate -> step -> validate ran the pre-step weights through it"""
        self.linear1.compile(device)
        self.linear2.compile(device)
        self._rowprog = None

    def _row_program(self, device):
        """Linear1 -> Gate -> Linear2 (+ x) as ONE row program (csrc/rowprog.hip; late r5: the node-level chain of a ConvBlock is launch-bound on small
        crystals -- three launches become one), or False (HG_ROWPROG=0 / HG_NODE_ROWPROG=0, "norm" activation, no kernel form, weights moving)"""
        if getattr(self, "_rowprog", None) is None:
            self._rowprog = False
            if (os.environ.get("HG_ROWPROG", "1") != "0" and os.environ.get("HG_NODE_ROWPROG", "1") != "0" and not getattr(self, "_rowprog_off", False)
                    and self.nonlinearity_type == "gate" and isinstance(self.linear1._dp, ops.DeviceLinear)):
                w = lambda m: m.weight.detach().cpu().double().numpy()
                li, lgi, lgo = P.PlanarLayout(self.irreps_in), P.PlanarLayout(self.gate_in), P.PlanarLayout(self.gate_out)
                try:
                    rp = P.build_row_program([("linear", P.o3_linear_mats(w(self.linear1), self.irreps_in, self.gate_in), li, lgi, False),
                                              ("gate", self._tab_np, lgi.dim, lgo.dim),
                                              ("linear", P.o3_linear_mats(w(self.linear2), self.gate_out, self.irreps_in), lgo, li, bool(self.resnet))], li.dim)
                    self._rowprog = ops.DeviceRowProgram(rp, device)
                except NotImplementedError:
                    pass
        return self._rowprog

    def _act(self, y1):
        return ops.norm_act(y1, self._tab) if self.nonlinearity_type == "norm" else ops.gate(y1, self._tab, self._cst)

    def _act_backward(self, y1, g_y2):
        return ops.norm_act_backward(y1, g_y2, self._tab) if self.nonlinearity_type == "norm" else ops.gate_backward(y1, g_y2, self._tab, self._cst)

    def forward(self, x_planar, extra=None, skip=None):
        """x + Lin2(Gate(Lin1(x))) [+ extra]  on planar rows.  skip = (a Linear, its input rows): extra = that Linear's result."""
        if self._tab is None:
            self.compile(x_planar.device)
        if skip is not None:                                   # (Linear, its input rows): `extra` = that Linear's result, evaluated where it is cheapest
            rp = self._row_program(x_planar.device)
            if rp:
                return skip[0](skip[1], res=[ops.row_program(rp, x_planar, tag="residual_block")])     # two launches: the chain, the skip Linear with the add in its epilogue
            extra = skip[0](skip[1])
        res = ([x_planar] if self.resnet else []) + ([extra] if extra is not None else [])
        return self.linear2(self._act(self.linear1(x_planar)), res=res)      # adds fused into linear2's epilogue


    def backward(self, x_planar, gy_planar, extra_given: bool = False):
        """gradient of forward(x, extra) = [x +] Lin2(Gate(Lin1(x))) [+ extra] for the output gradient gy (planar rows): returns
        (g_x, {"linear1.weight": ..., "linear2.weight": ...}) -- and g_extra = gy.  Recomputes the two cheap intermediates; the data
        gradients run on hg_linear_planar (transposed blocks) and hg_gate_backward, the weight gradients are one GEMM per path."""
        if self._tab is None:
            self.compile(x_planar.device)
        y1 = self.linear1(x_planar)                            # gate input rows
        y2 = self._act(y1)
        g_w2 = self.linear2.weight_grad(y2, gy_planar)
        g_y2 = self.linear2.backward_data(gy_planar)
        g_y1 = self._act_backward(y1, g_y2)
        g_w1 = self.linear1.weight_grad(x_planar, g_y1)
        g_x = self.linear1.backward_data(g_y1)
        if self.resnet:
            g_x = g_x + gy_planar
        return g_x, {"linear1.weight": g_w1, "linear2.weight": g_w2}


class ConvBlockE3(nn.Module):
    def __init__(self, irreps, irreps_sh, num_radial, radial_MLP, lite_mode=False):
        super().__init__()
        self.residual = ResidualBlock(irreps, irreps)
        self.conv_tp = MessagePackBlock(irreps, irreps, irreps_sh, irreps, num_radial, radial_MLP, lite_mode)
        self.skip_linear = E3Linear(irreps, irreps)

    def compile(self, device):
        self.residual.compile(device)
        self.conv_tp.compile(device, unrotate=True)
        self.skip_linear.compile(device)


class AttentionBlockE3(nn.Module):
    """AttentionBlockE3 of the reference (hamgnn/nn/attention.py:167-360), parameter names included.  `linear_query` and `max_radius`
    exist in the reference's state_dict but its forward never reads them (:339-340 use `linear_key` for key AND query): they are kept
    as parameter slots so that checkpoints load verified."""

    def __init__(self, irreps, irreps_sh, num_radial, num_heads, max_radius, radial_MLP):
        super().__init__()
        self.irreps, self.num_heads, self.cutoff = Irreps(irreps), int(num_heads), float(max_radius)
        self._head_tab_np, self.head_dim = P.attention_head_table(self.irreps, self.num_heads)
        self.register_buffer("max_radius", torch.tensor(float(max_radius)))
        self.cutoff_func = nn.Module()
        self.cutoff_func.cut_param = nn.Parameter(torch.tensor(10.0))                 # SoftUnitStepCutoff (utils/cutoff_functions.py:82)
        self.linear_up_src, self.linear_up_tar, self.linear_up_edge = E3Linear(irreps, irreps), E3Linear(irreps, irreps), E3Linear(irreps, irreps)
        self.residual = ResidualBlock(irreps, irreps)
        self.conv_tp_value = MessagePackBlock(irreps, irreps, irreps_sh, irreps, num_radial, radial_MLP)
        self.linear_key, self.linear_query = E3Linear(irreps, irreps), E3Linear(irreps, irreps)
        self.skip_linear = E3Linear(irreps, irreps)

    def compile(self, device):
        for m in (self.linear_up_src, self.linear_up_tar, self.linear_up_edge, self.residual, self.linear_key, self.skip_linear):
            m.compile(device)
        self.conv_tp_value.compile(device, unrotate=True)      # values leave the kernel in the global frame, like ConvBlockE3's messages
        self._head_tab = torch.from_numpy(self._head_tab_np).to(device)
        self._cut = self.cutoff_func.cut_param.detach().float().reshape(1).contiguous().to(device)

    def refresh(self, device):
        """after an optimiser step: the Linear tables and the cutoff parameter on the host (< 1 ms each), the value block's programs on
        the device (hamgnn_amd/repack.py)"""
        for m in (self.linear_up_src, self.linear_up_tar, self.linear_up_edge, self.linear_key, self.skip_linear):
            m.compile(device)
        self.residual.refresh(device)
        if not self.conv_tp_value.refresh():
            self.conv_tp_value.compile(device, unrotate=True)
        self._cut = self.cutoff_func.cut_param.detach().float().reshape(1).contiguous().to(device)

    def run(self, node, f, geo: ops.Geometry, rot_tab, rowptr, perm, data=None, structural_zeros: bool = False):
        """node [N, Dp] planar (global frame), f [E, Dp] planar edge features (edge frame) -> new node rows (attention.py:315-360).
        data: the graph, for edge-sharded runs (the soft-max of a node then spans the edges of several ranks); structural_zeros: see MessagePackBlock._dp_for"""
        from . import parallel
        sc = self.skip_linear(node)
        K = self.linear_key(node)
        value = self.conv_tp_value.run_nodes(self.linear_up_src(node), self.linear_up_tar(node), self.linear_up_edge(f), geo, rot_tab, structural_zeros=structural_zeros)
        agg = ops.attention_aggregate(K, value, geo, rowptr, perm, self._head_tab, self.num_heads, self.head_dim, self._cut, self.cutoff)
        if data is not None and parallel.is_sharded(data):
            agg = self._merge_sharded_softmax(agg, K, geo)
        return self.residual(agg, extra=sc)

    def _merge_sharded_softmax(self, agg_local, K, geo):
        """edge-sharded attention: `agg_local` is normalised over THIS rank's incoming edges.  With the rank's soft-max statistics per
        (node, head) -- m_r = max logit, Z_r = sum exp(logit - m_r) -- the rank's un-normalised sum is agg_local (Z_r + 1e-16); the global
        result is  sum_r exp(m_r - m) (un-normalised sum)_r / (sum_r exp(m_r - m) Z_r + 1e-16),  m = max_r m_r:  one MAX all-reduce of
        [N, H], one SUM all-reduce of [N, H] and one of [N, Dp] per block (RCCL)."""
        import torch.distributed as dist
        N, H = K.shape[0], self.num_heads
        logits = ops.attention_logits(K, geo, self._head_tab, H, self.head_dim, self._cut, self.cutoff)
        dst = geo.dst.long()
        m_r = torch.full((N, H), -float("inf"), device=K.device, dtype=logits.dtype).scatter_reduce(0, dst[:, None].expand(-1, H), logits, "amax")
        Z_r = ops.scatter_rows(dst, torch.exp(logits - m_r[dst]), N)
        m = m_r.clone()
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        scale = torch.where(torch.isinf(m_r), torch.zeros_like(m_r), torch.exp(m_r - m))
        Zs = Z_r * scale
        dist.all_reduce(Zs, op=dist.ReduceOp.SUM)
        M = torch.zeros(K.shape[1], H, device=K.device, dtype=agg_local.dtype)   # column -> head indicator (padding columns: zero)
        cols = torch.nonzero(self._head_tab >= 0).reshape(-1)
        M[cols, self._head_tab[cols].long()] = 1.0
        num = agg_local * (((Z_r + 1e-16) * scale) @ M.t())
        dist.all_reduce(num, op=dist.ReduceOp.SUM)
        return (num / ((Zs + 1e-16) @ M.t()).clamp_min(1e-30)).contiguous()


    def backward(self, node, f, geo: ops.Geometry, rot_tab, topo, g_out, chunk: int = 65536, data=None):
        """gradient of run(node, f, ...) for the gradient g_out of the node rows it returned: (g_node, g_f (edge frame), {parameter name:
        gradient}).  data: the graph, for edge-sharded runs (soft-max statistics and node-level sums then span the ranks).  ResidualBlock / Linears: the streaming-kernel adjoints; the attention aggregation (soft-max over incoming edges, the
        learnable soft cutoff): hamgnn_amd/backward_attn.py; the value MessagePackBlock: its adjoint / materialisation programs, with the
        sender / receiver segment sums for the two gathered node inputs."""
        from .backward_attn import attention_backward
        N = node.shape[0]
        K = self.linear_key(node)
        us, ut, ue = self.linear_up_src(node), self.linear_up_tar(node), self.linear_up_edge(f)
        value = self.conv_tp_value.run_nodes(us, ut, ue, geo, rot_tab)                 # [E, Dp], global frame
        rowptr, perm = topo.receiver_csr()
        agg = ops.attention_aggregate(K, value, geo, rowptr, perm, self._head_tab, self.num_heads, self.head_dim, self._cut, self.cutoff)
        from . import parallel
        allreduce = None
        if data is not None and parallel.is_sharded(data):
            import torch.distributed as dist
            agg = self._merge_sharded_softmax(agg, K, geo)
            allreduce = lambda t, op: dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        grads = {}
        g_agg, g_res = self.residual.backward(agg, g_out, extra_given=True)
        grads.update({"residual." + k: v for k, v in g_res.items()})
        grads["skip_linear.weight"] = self.skip_linear.weight_grad(node, g_out)
        g_node = self.skip_linear.backward_data(g_out)
        g_K, g_V, g_p = attention_backward(K, value, g_agg, geo.src, geo.dst, geo.length, self._head_tab, self.num_heads, self.head_dim,
                                           self._cut, self.cutoff, allreduce=allreduce)
        grads["cutoff_func.cut_param"] = g_p.reshape(())
        grads["linear_key.weight"] = self.linear_key.weight_grad(node, g_K)
        grads["linear_query.weight"] = torch.zeros_like(self.linear_query.weight)     # a parameter the reference's forward never reads
        g_node = g_node + self.linear_key.backward_data(g_K.contiguous())
        g_V = g_V.contiguous()
        gs, gd, ge, g_cv = self.conv_tp_value.backward(us, ut, ue, geo, rot_tab, g_V, out_is_global=True, chunk=chunk)
        grads.update({"conv_tp_value." + k: v for k, v in g_cv.items()})
        g_us = ops.segment_sum(gs, *topo.sender_csr(), N)
        g_ut = ops.segment_sum(gd, rowptr, perm, N)
        if allreduce is not None:                              # this rank's edges -> all edges
            allreduce(g_us, "sum")
            allreduce(g_ut, "sum")
        grads["linear_up_src.weight"] = self.linear_up_src.weight_grad(node, g_us)
        grads["linear_up_tar.weight"] = self.linear_up_tar.weight_grad(node, g_ut)
        grads["linear_up_edge.weight"] = self.linear_up_edge.weight_grad(f, ge)
        g_node = g_node + self.linear_up_src.backward_data(g_us) + self.linear_up_tar.backward_data(g_ut)
        return g_node, self.linear_up_edge.backward_data(ge), grads


class PairInteractionBlock(nn.Module):
    def __init__(self, irreps, irreps_sh, num_radial, radial_MLP, use_skip_connections=True, legacy_edge_update=False, lite_mode=False):
        super().__init__()
        self.use_skip_connections, self.legacy_edge_update, self.lite_mode = use_skip_connections, legacy_edge_update, lite_mode
        self.linear_up_src = E3Linear(irreps, irreps)
        self.linear_up_tar = E3Linear(irreps, irreps)
        self.conv_tp = MessagePackBlock(irreps, irreps, irreps_sh, irreps, num_radial, radial_MLP, lite_mode)
        if use_skip_connections:
            self.skip_linear = E3Linear(irreps, irreps)

    def compile(self, device):
        self._up_both = None
        self.linear_up_src.compile(device)
        self.linear_up_tar.compile(device)
        skip = self.skip_linear.weight.detach().cpu().double().numpy() if self.use_skip_connections else None
        if self.lite_mode:                                     # the combine post-op must not touch the skip term: separate launch + add
            self.conv_tp.compile(device, unrotate=False)
            if self.use_skip_connections:
                self.skip_linear.compile(device)
        else:
            self.conv_tp._skip_source = (self.skip_linear,) if self.use_skip_connections else None   # (a tuple: not registered as a sub-module)
            self.conv_tp.compile(device, unrotate=False, skip_weight=skip)   # skip o3.Linear fused as extra items
            if self.use_skip_connections:
                self.skip_linear._stale = True                 # its forward is fused above; only the backward uses the module's own (adjoint) tables


    def linear_up_both(self, node):
        """(linear_up_src(node), linear_up_tar(node)) as ONE launch: an o3.Linear into the doubled irreps, the two results are the halves of its rows
        (views with the doubled row stride -- the edge kernel gathers node rows by pointer + stride).  Inference only: built from the weights of the moment,
        dropped by compile() / refresh() (late r5: small crystals are launch-bound)."""
        if getattr(self, "_up_both", None) is None:
            self._up_both = False
            if os.environ.get("HG_NODE_ROWPROG", "1") != "0" and isinstance(self.linear_up_src._dp, ops.DeviceLinear):
                irr = self.linear_up_src.irreps_in
                w = lambda m: m.weight.detach().cpu().double().numpy()
                ms, mt = P.o3_linear_mats(w(self.linear_up_src), irr, irr), P.o3_linear_mats(w(self.linear_up_tar), irr, irr)
                K = len(irr)
                mats = dict(ms)
                mats.update({(i, K + k): M for (i, k), M in mt.items()})
                both = Irreps(list(irr) + list(irr))
                self._up_both = ops.DeviceLinear(P.linear_tables(mats, P.PlanarLayout(irr), P.PlanarLayout(both)), node.device)
        if not self._up_both:
            return self.linear_up_src(node), self.linear_up_tar(node)
        y = ops.linear_planar(self._up_both, node)
        half = y.shape[1] // 2
        return y[:, :half], y[:, half:]

    def refresh(self, device):
        """after an optimiser step: the two linear_up tables (host, < 1 ms) and the message block's programs (device repack)"""
        self._up_both = None
        self.linear_up_src.compile(device)
        self.linear_up_tar.compile(device)
        skip = self.skip_linear.weight if (self.use_skip_connections and not self.lite_mode) else None
        if not self.conv_tp.refresh(skip=skip):
            self.compile(device)
        elif self.use_skip_connections:
            self.skip_linear._stale = True


class _EmbTP(nn.Module):
    """TensorProductWithMemoryOptimizationWithWeight parameter holder (tensor_products.py:51-189)."""

    def __init__(self, irreps_in, irreps_sh, irreps_out, num_radial, radial_MLP, lite_mode=False):
        super().__init__()
        if lite_mode:                                          # uvu, no TP weights; mid multiplicity = input multiplicity
            self.linear_scaler = nn.Module()
            self.linear_scaler.linear_out = _MidLinear(Irreps(irreps_in), Irreps(irreps_sh), Irreps(irreps_out))
            ins = P.tp_instructions(Irreps(irreps_in), Irreps(irreps_sh), Irreps(irreps_out))
            self.linear_scaler.weight_numel = sum(Irreps(irreps_in)[i][0] for i, _, _, _ in ins)
        else:
            self.tensor_product = E3TensorProduct(irreps_in, irreps_sh, irreps_out)
            self.linear_scaler = LinearScaleWithWeights(self.tensor_product, Irreps(irreps_out))
        self.weight_generator = FullyConnectedNet([num_radial] + list(radial_MLP) + [self.linear_scaler.weight_numel])


class PairInteractionEmbeddingBlock(nn.Module):
    def __init__(self, num_types, irreps_sh, irreps_out, num_radial, radial_MLP, lite_mode=False):
        super().__init__()
        self.num_types, self.lite_mode = num_types, lite_mode
        attrs = Irreps([(num_types, 0, 1)])
        self.irreps_sh, self.irreps_out = Irreps(irreps_sh), Irreps(irreps_out)
        self.linear_up_src = E3Linear(attrs, attrs)
        self.linear_up_dst = E3Linear(attrs, attrs)
        self.conv_tp = _EmbTP(attrs, self.irreps_sh, self.irreps_out, num_radial, radial_MLP, lite_mode)

    def compile(self, device):
        T = self.num_types
        s = 1.0 / math.sqrt(T)
        self._Ts = (self.linear_up_src.weight.detach().double().reshape(T, T) * s).float().contiguous().to(device)
        self._Td = (self.linear_up_dst.weight.detach().double().reshape(T, T) * s).float().contiguous().to(device)
        # the block's weights on the host, read HERE -- at the start of a step, when the device is idle -- and kept for backward(): read again
        # there, the device -> host copy drained the whole forward + head backward queue in the middle of the step (61 of 155 ms on Si-512)
        self._sd_np = _np_sd(self.conv_tp)
        self._dp = ops.DeviceProgram(P.build_embedding_program(self._sd_np, T, self.irreps_sh, self.irreps_out, self.lite_mode), device)
        self._h = self.conv_tp.weight_generator.hidden_layers(device)
        self._Tp = P.PlanarLayout([(T, 0, 1)]).dim
        self._wgrad_prev, self._wgrad = (getattr(self, "_wgrad", None) or getattr(self, "_wgrad_prev", None)), None

    def backward(self, z, geo: ops.Geometry, g_f, chunk: int = 65536, delta=None):
        """gradients of every parameter of the block for the gradient g_f of the edge rows it returned (planar, edge frame):
        conv_tp.* through the materialisation programs (backward_mp), linear_up_src / linear_up_dst from the gradient of the
        num_types scalar input channels (an index_add over the element of the sender / receiver).
        delta: the charge-doping correction the forward ran with ([N, num_types]; attrs = one_hot(z) + delta): its gradient is returned
        under the key "_g_delta" (the caller backpropagates it through the charge MLP)."""
        from . import backward_mp as BM
        dev, T = g_f.device, self.num_types
        if self._wgrad is None:
            sd = getattr(self, "_sd_np", None) or _np_sd(self.conv_tp)
            wg = BM.TPWeightGrad(sd, P.embedding_wgrad_branches(sd, T, self.lite_mode), self.irreps_sh, self.irreps_out)
            wg.adopt_constants(self._wgrad_prev[0] if getattr(self, "_wgrad_prev", None) else None)
            self._wgrad = [wg, None, None]                     # (the two materialisation programs are uploaded only if the fused route below is not available)
        wg, dpA, dpB = self._wgrad
        if delta is not None:
            Ts, Td = (self._Ts[z] + delta @ self._Ts).contiguous(), (self._Td[z] + delta @ self._Td).contiguous()
            x = ops.embed_lookup(Ts, Td, torch.arange(z.shape[0], device=dev), geo.src, geo.dst, geo.E, T, self._Tp)
        else:
            x = ops.embed_lookup(self._Ts, self._Td, z, geo.src, geo.dst, geo.E, T, self._Tp)
        cst = float(P.ACT_CONSTS[P.ACT_SILU])
        fused = self._fused_backward_tables(wg, dev)
        if fused is not None:
            # late r5: the embedding TP's weight gradients on the fused kernel (its 96-channel 0e row as two 48-channel sources) and its input gradient as an
            # adjoint program on the forward's kernels, instead of the materialisation programs + per-group GEMMs (4.5 -> 1.5 ms of a Si-512 step)
            dwf, dp_adj = fused
            h = ops.radial_hidden_cached(geo, self._h, cst)
            run = lambda srcs, g_, hn, he: ops.tp_wgrad(dwf, srcs, g_, hn, he)
            grads = BM.tp_weight_grads_fused(wg, dwf, run, [x[:, :T // 2], x[:, T // 2:T]], g_f, geo.rbf, cst, hidden={"emb": h})
            gx = [ops.tp_fused(dp_adj, [g_f], geo.E, h, None, geo, tag="embedding_adjoint")]
        else:
            if dpA is None:
                dpA, dpB = ops.DeviceProgram(wg.progA, dev, schedule="seg"), ops.DeviceProgram(wg.progB, dev, schedule="seg")
                self._wgrad[1:] = [dpA, dpB]
            grads, gx = BM.tp_weight_grads(wg, _wgrad_runner(wg, dpA, dpB), [x], g_f, geo.rbf, cst, chunk=chunk, want_gx=True)
        out = {"conv_tp." + k: v for k, v in grads.items()}
        gx = gx[0][:, :T]
        s = 1.0 / math.sqrt(T)
        N = z.shape[0]
        g_delta = torch.zeros(N, T, device=dev, dtype=gx.dtype) if delta is not None else None
        for name, idx, tab in (("linear_up_src", geo.src, self._Ts), ("linear_up_dst", geo.dst, self._Td)):
            g_atom = ops.scatter_rows(idx, gx, N)              # gradient of the per-atom table rows (fixed summation order: no float atomics)
            gT = ops.scatter_rows(z, g_atom, T)                # rows one_hot(z) @ table
            if delta is not None:                              # rows (one_hot(z) + delta) @ table
                gT = gT + delta.t() @ g_atom
                g_delta += g_atom @ tab.t()
            out[name + ".weight"] = (gT * s).reshape(-1)
        if g_delta is not None:
            out["_g_delta"] = g_delta
        return out

    def _fused_backward_tables(self, wg, dev):
        """(fused weight-gradient tables, adjoint program) of the embedding TP on the device, or None: lite_mode (no TP weights), HG_WGRAD=rows, a radial MLP
        that is not 64 wide, num_types not a multiple of 8 -- the materialisation route then.  Rebuilt when compile() replaced the weights."""
        if self.lite_mode or os.environ.get("HG_WGRAD", "fused") != "fused":
            return None
        cur = getattr(self, "_fused_bw", None)
        if cur is None or cur[0] is not wg:
            sd = wg.sd
            try:
                wf = P.build_tp_wgrad_fused(P.embedding_wgrad_branches_split(sd, self.num_types), self.irreps_sh, self.irreps_out, wg.H)
                prog = P.build_embedding_adjoint_program(sd, self.num_types, self.irreps_sh, self.irreps_out)
                try:
                    dp = ops.DeviceProgram(prog, dev, schedule="is_parts" if os.environ.get("HG_MP_KERNEL", MP_KERNEL_DEFAULT) != "seg" else "seg")
                except NotImplementedError:
                    dp = ops.DeviceProgram(prog, dev, schedule="seg")
                cur = (wg, ops.DeviceWgFused(wf, dev), dp)
            except NotImplementedError:
                cur = (wg, None, None)
            self._fused_bw = cur
        return None if cur[1] is None else (cur[1], cur[2])

    def run(self, z, geo: ops.Geometry, delta=None):
        """delta: optional [N, num_types] charge-doping correction of the node attributes -> per-atom source / target tables"""
        if delta is not None:
            Ts, Td = (self._Ts[z] + delta @ self._Ts).contiguous(), (self._Td[z] + delta @ self._Td).contiguous()
            z = torch.arange(z.shape[0], device=z.device)
            x = ops.embed_lookup(Ts, Td, z, geo.src, geo.dst, geo.E, self.num_types, self._Tp)
        else:
            x = ops.embed_lookup(self._Ts, self._Td, z, geo.src, geo.dst, geo.E, self.num_types, self._Tp)
        h = ops.radial_hidden_cached(geo, self._h, float(P.ACT_CONSTS[P.ACT_SILU]))
        return ops.tp_fused(self._dp, [x], geo.E, h, None, geo, tag="embedding")      # edge features, edge-aligned frame


class ChargeEmbedding(nn.Module):
    """Embedding_block_q of the reference (hamgnn/toolbox/nequip/nn/embedding/_embedding_block.py:56-137), parameter names included
    (`mlp_q.fcs.0.0.{weight,bias}`, `mlp_q.fc_out.{weight,bias}`): the charge-dependent CORRECTION of the one-hot node attributes,
        delta[n] = mlp_q(gauss(q_n)) - mlp_q(gauss(0)),      node_attrs = one_hot(z) + delta.
    delta is an [N, num_types] node-level quantity (usually one charge per crystal): F = 8 Gaussian features through an 8 -> 8 -> num_types
    MLP.  It is evaluated with torch tensor ops on the device (a few kFLOP per atom, once per forward); everything that consumes it stays
    on the HIP kernels: the embedding look-ups become per-ATOM tables `table[z] + delta @ table`."""

    def __init__(self, num_types, num_charge_attr_feas=8):
        super().__init__()
        F = int(num_charge_attr_feas)
        self.charge_min, self.charge_max = -8.0, 8.0
        width = (self.charge_max - self.charge_min) / (F - 1) if F > 1 else 1.0
        centers = torch.linspace(self.charge_min, self.charge_max, steps=F)
        self.register_buffer("charge_centers", centers)
        self.register_buffer("charge_gamma", torch.tensor(1.0 / width ** 2))
        self.register_buffer("neutral_charge_attrs", torch.exp(-(1.0 / width ** 2) * centers * centers).view(1, -1))
        self.mlp_q = nn.Module()
        self.mlp_q.fcs = nn.ModuleList([nn.Sequential(nn.Linear(F, F, bias=True), nn.Softplus())])
        self.mlp_q.fc_out = nn.Linear(F, num_types)

    def _mlp(self, x):
        for fc in self.mlp_q.fcs:
            x = fc(x)
        return self.mlp_q.fc_out(x)

    def delta(self, doping_charge, batch, N, device):
        q = torch.as_tensor(doping_charge, dtype=torch.float32, device=device)
        q = q.view(1) if q.dim() == 0 else q
        q = q.view(-1, 1) if q.dim() == 1 else q
        if batch is not None and q.size(0) != N:
            q = q[batch.view(-1)]
        elif q.size(0) != N:
            q = q[:1].expand(N, -1)
        q = q.clamp(self.charge_min, self.charge_max)
        diff = q - self.charge_centers.view(1, -1)
        attrs = torch.exp(-self.charge_gamma * diff * diff)
        return self._mlp(attrs) - self._mlp(self.neutral_charge_attrs.expand(attrs.size(0), -1))


class HamLayer(nn.Module):
    def __init__(self, irreps_in, ham_irreps: Irreps, keep=None, nonlinearity_type="gate"):
        """keep: optional bool per output irrep -- outputs the caller never reads (they keep their weights, for checkpoint
        compatibility, but no GEMM rows are spent on them; SOC/su2 head).  nonlinearity_type: of the ResidualBlock (hamgnn_output.py:38-58)."""
        super().__init__()
        self.irreps_in, self.ham_irreps, self.keep = Irreps(irreps_in), ham_irreps, keep
        self.residual_block = ResidualBlock(irreps_in, irreps_in, nonlinearity_type=nonlinearity_type)
        self.linear_transform = E3Linear(irreps_in, ham_irreps)

    def input_irreps_read(self):
        """the (l, parity) classes of the INPUT rows this network's result depends on.  Every map of the chain connects equal (l, p) only -- o3.Linear by
        construction, the Gate multiplies an irrep by a 0e gate scalar, NormActivation rescales an irrep by a function of its own norm (hamgnn_output.py:38-58;
        e3nn nn/_gate.py, nn/_normact.py) -- so an output irrep (L, p) sees the input blocks (L, p) and, through the gates, (0, +1); nothing else.
        (tests: perturbing any other input block leaves the result bit-identical.)"""
        keep = self.keep if self.keep is not None else [True] * len(self.ham_irreps)
        return frozenset((int(l), int(p)) for (m, l, p), kp in zip(self.ham_irreps, keep) if kp and m > 0) | {(0, 1)}

    def compile(self, device):
        self.residual_block.compile(device)
        self._dp_adj = None
        self._rowprog = None                                   # the fused chain (csrc/rowprog.hip) is built on first use, from the weights of that moment
        self._rowprog_off = getattr(self, "_rowprog_off", False)    # set by training._invalidate: separate kernels while the weights move
        W = self.linear_transform.weight.detach().cpu().double().numpy()
        self._W_np = W                                         # kept for backward(): a device -> host copy there would drain the queue mid-step
        stream = os.environ.get("HG_LINEAR_KERNEL", "stream") != "seg"
        if all(m == 1 for m, _, _ in self.ham_irreps):         # hamiltonian irreps: regroup the multiplicity-1 outputs by (L,p)
            if stream:
                mats, self.girr, self.slot_pos = P.ham_linear_mats(W, self.irreps_in, self.ham_irreps, self.keep)
                self._dp = ops.DeviceLinear(P.linear_tables(mats, P.PlanarLayout(self.irreps_in), P.PlanarLayout(self.girr)), device)
                return
            prog, self.girr, self.slot_pos = P.build_ham_linear_program(W, self.irreps_in, self.ham_irreps, self.keep)
        else:                                                  # xi networks (nao^2 x 0e): a plain o3.Linear
            self.girr, self.slot_pos = self.ham_irreps, None
            if stream:
                self._dp = ops.DeviceLinear(P.build_linear_tables(W, self.irreps_in, self.ham_irreps), device)
                return
            prog = P.build_linear_program(W, self.irreps_in, self.ham_irreps)
        self._dp = ops.DeviceProgram(prog, device)

    def _slot_gather(self, girr, slot_pos, dev):
        """(index into [flat weight gradient of o3.Linear(irreps_in -> girr) | 0], scale) that give linear_transform.weight's gradient: the reference's
        paths (i_in, slot) are columns of the (i_in, group) blocks, normalised by the slot's fan-in instead of the group's.  Structure only: cached."""
        cur = getattr(self, "_slot_gather_tab", None)
        if cur is None or cur[0] != str(dev):
            girr = Irreps(girr)
            paths = [(i, g) for i, (_, l1, p1) in enumerate(self.irreps_in) for g, (_, l2, p2) in enumerate(girr) if (l1, p1) == (l2, p2)]
            off, fan_g, o = {}, {}, 0
            for i, g in paths:
                off[(i, g)] = o
                o += self.irreps_in[i][0] * girr[g][0]
                fan_g[g] = fan_g.get(g, 0) + self.irreps_in[i][0]
            fan = {}
            for i, (mi, l1, p1) in enumerate(self.irreps_in):
                for s_, (_, L, p) in enumerate(self.ham_irreps):
                    if (l1, p1) == (L, p):
                        fan[s_] = fan.get(s_, 0) + mi
            idx, scale = [], []
            for i, (mi, l1, p1) in enumerate(self.irreps_in):
                for s_, (_, L, p) in enumerate(self.ham_irreps):
                    if (l1, p1) != (L, p):
                        continue
                    pos = slot_pos[s_]
                    for u in range(mi):
                        if pos is None:                          # an output the caller never reads (keep): zero gradient
                            idx.append(o)
                            scale.append(0.0)
                        else:
                            idx.append(off[(i, pos[0])] + u * girr[pos[0]][0] + pos[1])
                            scale.append(math.sqrt(fan_g[pos[0]] / fan[s_]))
            cur = (str(dev), torch.tensor(idx, dtype=torch.int64, device=dev), torch.tensor(scale, dtype=torch.float32, device=dev))
            self._slot_gather_tab = cur
        return cur[1], cur[2]

    # ---- backward
    def backward(self, x_planar, g_out_planar):
        """gradient of forward(x) = linear_transform(residual_block(x)) for the gradient of its (grouped planar) output rows: returns
        (g_x, {parameter name: gradient in the reference's flat layout})"""
        if not isinstance(self._dp, ops.DeviceLinear):
            raise NotImplementedError("HamLayer.backward: streaming Linear path only (HG_LINEAR_KERNEL=seg has no adjoint tables)")
        W_host = lambda: self._W_np if getattr(self, "_W_np", None) is not None else self.linear_transform.weight.detach().cpu().double().numpy()
        if self.slot_pos is None:                               # xi networks: a plain o3.Linear (e.g. irreps_in -> nao^2 x 0e)
            W = W_host()
            if getattr(self, "_dp_adj", None) is None:
                self._dp_adj = ops.DeviceLinear(P.build_linear_adjoint_tables(W, self.irreps_in, self.ham_irreps), x_planar.device)
            y = self.residual_block(x_planar)
            g_y = ops.linear_planar(self._dp_adj, g_out_planar, tag="linear_adjoint")
            g_x, g_res = self.residual_block.backward(x_planar, g_y)
            grads = {"linear_transform.weight": o3_linear_weight_grad(self.irreps_in, self.ham_irreps, y, g_out_planar)}
            grads.update({"residual_block." + k: v for k, v in g_res.items()})
            return g_x, grads
        girr, slot_pos = self.girr, self.slot_pos              # (structure: as compile() found it)
        dev = x_planar.device
        if getattr(self, "_dp_adj", None) is None:
            mats = P.ham_linear_mats(W_host(), self.irreps_in, self.ham_irreps, self.keep)[0]
            self._dp_adj = ops.DeviceLinear(P.linear_tables({(g, i): M.T for (i, g), M in mats.items()}, P.PlanarLayout(girr),
                                                            P.PlanarLayout(self.irreps_in)), dev)
        y = self.residual_block(x_planar)
        # weight gradient of linear_transform in e3nn's flat layout: for i_in, for i_out (matching ir): block (mul_in, 1) / sqrt(fan_in).  The slots of one
        # (L, p) are the channels of a group irrep: ONE o3.Linear(irreps_in -> girr) weight gradient (hg_linear_wgrad: all paths in one launch) and a
        # gather with the slots' own normalisation (r5; before: one GEMM on strided copies per path and one slice per slot -- ~500 launches per network)
        flat = o3_linear_weight_grad(self.irreps_in, girr, y, g_out_planar)
        idx, scale = self._slot_gather(girr, slot_pos, dev)
        gw = [torch.cat([flat, flat.new_zeros(1)])[idx] * scale]
        g_y = ops.linear_planar(self._dp_adj, g_out_planar, tag="linear_adjoint")
        g_x, g_res = self.residual_block.backward(x_planar, g_y)
        grads = {"linear_transform.weight": torch.cat(gw)}
        grads.update({"residual_block." + k: v for k, v in g_res.items()})
        return g_x, grads

    def _row_program(self, device):
        """Linear1 -> Gate -> Linear2 (+ x) -> linear_transform as ONE row program (plan.build_row_program), or False when the chain has no
        kernel form (HG_ROWPROG=0, channel counts beyond the kernel's unit shape, LDS)"""
        if getattr(self, "_rowprog", None) is None:
            self._rowprog = False
            if (os.environ.get("HG_ROWPROG", "1") != "0" and not getattr(self, "_rowprog_off", False) and isinstance(self._dp, ops.DeviceLinear)
                    and self.residual_block.nonlinearity_type == "gate"):      # (the row program has a gate stage only: "norm" runs Linear / hg_norm_act / Linear)
                rb = self.residual_block
                w = lambda m: m.weight.detach().cpu().double().numpy()
                li, lgi, lgo = P.PlanarLayout(self.irreps_in), P.PlanarLayout(rb.gate_in), P.PlanarLayout(rb.gate_out)
                if self.slot_pos is not None:
                    m3 = P.ham_linear_mats(w(self.linear_transform), self.irreps_in, self.ham_irreps, self.keep)[0]
                else:
                    m3 = P.o3_linear_mats(w(self.linear_transform), self.irreps_in, self.ham_irreps)
                try:
                    rp = P.build_row_program([("linear", P.o3_linear_mats(w(rb.linear1), self.irreps_in, rb.gate_in), li, lgi, False),
                                              ("gate", rb._tab_np, lgi.dim, lgo.dim),
                                              ("linear", P.o3_linear_mats(w(rb.linear2), rb.gate_out, self.irreps_in), lgo, li, bool(rb.resnet)),
                                              ("linear", m3, li, P.PlanarLayout(self.girr), False)], li.dim)
                    self._rowprog = ops.DeviceRowProgram(rp, device)
                except NotImplementedError:
                    pass
        return self._rowprog

    def forward(self, x_planar):
        rp = self._row_program(x_planar.device)
        if rp:                                                 # one pass: the row is read once, the coefficient row written once
            return ops.row_program(rp, x_planar, tag="ham_layer")
        y = self.residual_block(x_planar)
        if isinstance(self._dp, ops.DeviceLinear):
            return ops.linear_planar(self._dp, y)                              # planar rows grouped by (L,p)
        return ops.tp_fused(self._dp, [y], y.shape[0])


# ------------------------------------------------------------------------------------------------ correlation product (a21)
class _Contraction(nn.Module):
    """parameter holder of one MACE Contraction (toolbox/mace/modules/symmetric_contraction.py:101-233): weights_max for nu =
    correlation, weights[0], weights[1], ... for nu = correlation - 1, ..., 1; shapes [num_elements, num_paths, num_features]."""

    def __init__(self, num_elements, ks, num_features):
        super().__init__()
        mk = lambda k: nn.Parameter(torch.randn(num_elements, k, num_features) / max(1, k))
        self.weights_max = mk(ks[-1])                          # ks: number of paths for nu = 1 ... correlation
        self.weights = nn.ParameterList([mk(k) for k in reversed(ks[:-1])])

    def by_nu(self, nu):
        corr = len(self.weights) + 1
        return self.weights_max if nu == corr else self.weights[corr - 1 - nu]

    def name_of(self, nu):
        corr = len(self.weights) + 1
        return "weights_max" if nu == corr else f"weights.{corr - 1 - nu}"


class _SymmetricContraction(nn.Module):
    def __init__(self, num_elements, Ks, num_features):
        super().__init__()
        self.contractions = nn.ModuleList([_Contraction(num_elements, ks, num_features) for ks in zip(*Ks)])


class _ProductBasis(nn.Module):
    def __init__(self, irreps_hidden, num_elements, Ks, num_features):
        super().__init__()
        self.symmetric_contractions = _SymmetricContraction(num_elements, Ks, num_features)
        self.linear = E3Linear(irreps_hidden, irreps_hidden)


class CorrProductBlock(nn.Module):
    """Drop-in for hamgnn/nn/interaction_blocks.py:168-260 (correlation 1, 2 -- the reference default -- or 3): linear_pre -> symmetric
    contraction with element-dependent weights -> prod.linear -> linear_out (+ linear_sc skip), all on planar node rows; same parameter
    names.  The nu <= 2 part of the contraction is the hg_sym_contraction kernel, the nu = 3 term hg_sym_contraction3 (backward: hamgnn_amd/corr3.py)."""

    def __init__(self, irreps_node_feats, num_hidden_features, correlation, num_elements, use_skip_connections=True):
        super().__init__()
        if correlation not in (1, 2, 3):
            raise NotImplementedError("CorrProductBlock: correlation 1, 2 (the reference default) and 3 are built")
        self.correlation = correlation
        self.irreps = Irreps(irreps_node_feats)
        if len({(l, p) for _, l, p in self.irreps}) != len(self.irreps):
            raise NotImplementedError("CorrProductBlock expects simplified node irreps (one entry per (l, p))")
        self.irreps_hidden = P.corr_hidden_irreps(self.irreps, num_hidden_features)
        self.num_hidden, self.num_elements, self.use_skip_connections = num_hidden_features, num_elements, use_skip_connections
        self._tab_np = P.sym_contraction_tables(self.irreps_hidden, correlation)
        self.linear_pre = E3Linear(self.irreps, self.irreps_hidden)
        self.linear_sc = E3Linear(self.irreps, self.irreps)
        Ks = [self._tab_np[f"K{nu}"] for nu in range(1, correlation + 1)]
        self.prod = _ProductBasis(self.irreps_hidden, num_elements, Ks, num_hidden_features)
        self.linear_out = E3Linear(self.irreps_hidden, self.irreps)
        self._tab = None

    def compile(self, device):
        for m in (self.linear_pre, self.linear_sc, self.prod.linear, self.linear_out):
            m.compile(device)
        d, cur = torch.device(device), (self._tab["ell_off"].device if self._tab is not None else None)
        if cur is None or cur.type != d.type or (d.index is not None and d.index != cur.index):   # structural: uploaded once per device, not per refresh
            self._tab = {k: (torch.from_numpy(v).to(device) if isinstance(v, np.ndarray) else v) for k, v in self._tab_np.items()}
        cons = self.prod.symmetric_contractions.contractions
        cat = lambda nu: torch.cat([c.by_nu(nu).detach() for c in cons], dim=1).float().contiguous().to(device)
        self._W1 = cat(1)
        # correlation 1: the kernel's nu = 2 loop runs over empty entry lists; it still wants a weight pointer
        self._W2 = cat(2) if self.correlation >= 2 else torch.zeros(self._W1.shape[0], 1, self._W1.shape[2], device=device)
        self._W3 = cat(3) if self.correlation >= 3 else None
        self._hdim = P.PlanarLayout(self.irreps_hidden).dim
        return self

    def _contract(self, h, zi, W1, W2, W3):
        c = ops.sym_contraction(h, zi, self.num_hidden, self._tab, W1, W2, self._hdim)
        if W3 is not None:                                     # correlation 3: its own kernel adds the nu = 3 term onto the same rows
            c = ops.sym_contraction3(h, zi, self.num_hidden, self._tab, W3, c)
        return c

    MIX_BYTES = 256 << 20          # per-node weight mixtures held at once (charge doping): [chunk, K_nu, C] floats over all nu

    def _mix_chunks(self, N):
        """node ranges whose per-node weight mixtures W[z_n] + delta_n @ W (all nu) stay below MIX_BYTES: with correlation 3 a node's blocks reach
        thousands of K x C floats, so N x that -- several times over in the backward -- is what an un-chunked pass would allocate"""
        per = 4 * sum(int(W[0].numel()) for W in (self._W1, self._W2, self._W3) if W is not None)
        step = max(64, int(self.MIX_BYTES // max(per, 1)))
        return [(a, min(N, a + step)) for a in range(0, N, step)]

    def _mixed(self, z, delta):
        """apply_charge_doping: the reference contracts the element weights with node_attrs = one_hot(z) + delta (interaction_blocks.py:251,
        symmetric_contraction.py einsum '...,ek'), i.e. every node gets its own mixture of the element blocks: W_eff[n] = W[z_n] + delta_n @ W.
        Returns (attrs [N, T], per-node weights of nu = 1, 2, 3, node index as the 'element' index) for the same kernel.  Callers pass node CHUNKS."""
        T = self._W1.shape[0]
        A = torch.nn.functional.one_hot(z.long(), T).to(self._W1.dtype) + delta.to(self._W1.dtype)
        mix = lambda W: None if W is None else (A @ W.reshape(T, -1)).reshape(-1, *W.shape[1:]).contiguous()
        return A, mix(self._W1), mix(self._W2), mix(self._W3), torch.arange(z.shape[0], device=z.device, dtype=z.dtype)

    def _contract_doped(self, h, z, delta):
        out = []
        for a, b in self._mix_chunks(int(z.shape[0])):
            _, W1, W2, W3, zi = self._mixed(z[a:b], delta[a:b])
            out.append(self._contract(h[a:b].contiguous(), zi, W1, W2, W3))
        return out[0] if len(out) == 1 else torch.cat(out, 0)

    def backward(self, node_planar, z, g_out, delta=None):
        """gradient of forward(node, z) for the gradient g_out of the rows it returned: (g_node, {parameter name: gradient}).
        Linears: streaming-kernel adjoints + GEMM weight gradients; the symmetric contraction: hamgnn_amd/backward_corr.py.
        With the charge-doping correction `delta` the gradient with respect to it comes back under the key "_g_delta"."""
        from .backward_corr import sym_contraction_backward
        if self._tab is None:
            self.compile(node_planar.device)
        h = self.linear_pre(node_planar)
        c = self._contract(h, z, self._W1, self._W2, self._W3) if delta is None else self._contract_doped(h, z, delta)
        p = self.prod.linear(c)
        grads = {"linear_out.weight": self.linear_out.weight_grad(p, g_out), }
        g_p = self.linear_out.backward_data(g_out)
        grads["prod.linear.weight"] = self.prod.linear.weight_grad(c, g_p)
        g_c = self.prod.linear.backward_data(g_p)

        def contraction_grads(hc, zi, W1, W2, W3, gc, per_node):
            g_h_, gW1, gW2 = sym_contraction_backward(self._tab, hc, zi, W1, W2, self.num_hidden, gc, per_node=per_node)
            gW_ = [gW1, gW2]
            if W3 is not None:
                from .corr3 import sym3_backward
                g_h3, gW3 = sym3_backward(self._tab, hc, zi, W3, self.num_hidden, gc, per_node=per_node)
                g_h_ = g_h_ + g_h3
                gW_.append(gW3)
            return g_h_, gW_[:self.correlation]

        if delta is None:
            g_h, gW = contraction_grads(h, z, self._W1, self._W2, self._W3, g_c, False)
        else:                                                  # per-node gradients back onto the element blocks and onto the attributes, chunk by chunk
            T = self._W1.shape[0]
            Wel = [self._W1, self._W2, self._W3][:self.correlation]
            g_h_parts, g_delta_parts, gW = [], [], [torch.zeros_like(W) for W in Wel]
            for a, b in self._mix_chunks(int(z.shape[0])):
                A, W1, W2, W3, zi = self._mixed(z[a:b], delta[a:b])
                g_h_c, gW_c = contraction_grads(h[a:b].contiguous(), zi, W1, W2, W3, g_c[a:b].contiguous(), True)
                flat = [g.reshape(g.shape[0], -1) for g in gW_c]
                g_h_parts.append(g_h_c)
                g_delta_parts.append(sum(f @ W.reshape(T, -1).t() for f, W in zip(flat, Wel)))
                for acc, f, W in zip(gW, flat, Wel):
                    acc += (A.t() @ f).reshape(W.shape)
            g_h = g_h_parts[0] if len(g_h_parts) == 1 else torch.cat(g_h_parts, 0)
            grads["_g_delta"] = g_delta_parts[0] if len(g_delta_parts) == 1 else torch.cat(g_delta_parts, 0)
        for nu, g in enumerate(gW, start=1):                   # the concatenated weights back to one block per target irrep
            k0 = 0
            for i, con in enumerate(self.prod.symmetric_contractions.contractions):
                n = con.by_nu(nu).shape[1]
                grads[f"prod.symmetric_contractions.contractions.{i}.{con.name_of(nu)}"] = g[:, k0:k0 + n]
                k0 += n
        grads["linear_pre.weight"] = self.linear_pre.weight_grad(node_planar, g_h)
        g_node = self.linear_pre.backward_data(g_h)
        if self.use_skip_connections:
            grads["linear_sc.weight"] = self.linear_sc.weight_grad(node_planar, g_out)
            g_node = g_node + self.linear_sc.backward_data(g_out)
        else:
            grads["linear_sc.weight"] = torch.zeros_like(self.linear_sc.weight).reshape(-1)
        return g_node, grads

    def forward(self, node_planar, z, delta=None):
        """returns the new planar node rows (the reference writes them back into the graph dict); delta: the charge-doping correction of
        the node attributes [N, num_elements] or None"""
        if self._tab is None:
            self.compile(node_planar.device)
        h = self.linear_pre(node_planar)
        if delta is not None:
            c = self._contract_doped(h, z, delta)
        else:
            c = self._contract(h, z, self._W1, self._W2, self._W3)
        skip = [self.linear_sc(node_planar)] if self.use_skip_connections else []
        return self.linear_out(self.prod.linear(c), res=skip)

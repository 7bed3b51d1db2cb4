"""The training step: what the reference's Lightning `training_step` + `loss.backward()` do per batch (hamgnn/main.py:389-420,
hamgnn/models/Model.py:150-196), on the HIP kernels and without an autograd graph.

  training_step(model, batch, ...)   forward with the layer inputs kept -> loss(es) -> head backward -> backbone backward -> `.grad` of EVERY
                                     parameter in the reference's names / flat layouts (any torch optimiser steps them); both backbones
                                     (HamGNNConvE3, HamGNNTransformer), the non-SOC / SOC so3 / SOC su2 heads, `losses=[{metric, prediction:
                                     hamiltonian | hamiltonian_real | hamiltonian_imag | band_energy, target, loss_weight}]` as in the reference's config; hamiltonian-type
                                     losses are multiplied by the head's `sparsity_ratio` as the reference does (Model.py:158-162)
  head_training_step(...)            the cheap variant for a frozen backbone (its representation can be reused across steps)
  allreduce_gradients(model)         data-parallel training (the reference's DDP): mean of the ranks' gradients, one flat bucket
  parallel.shard_graph(g, r, w)      model-parallel training of ONE large crystal: pass the rank's shard to training_step -- node-level partial
                                     sums are all-reduced inside the backward, per-edge parameter gradients summed over the ranks
  weights_changed(model)             after `optimizer.step()`: message blocks repack their weights on the device at the next forward
                                     (hamgnn_amd/repack.py); training_step calls it for the step that follows

No Lightning loop, no scheduler / logging / checkpoint policy: the caller owns those (`Model.save_checkpoint` writes the reference's layout)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import parallel
from .topo import gget


def _loss_and_grad(pred: torch.Tensor, target: torch.Tensor, metric: str):
    diff = pred - target
    n = diff.numel()
    metric = metric.lower()
    if metric == "mae":
        return diff.abs().mean(), torch.sign(diff) / n
    if metric == "mse":
        return (diff * diff).mean(), 2.0 * diff / n
    if metric == "rmse":
        rm = torch.sqrt((diff * diff).mean())
        return rm, diff / (n * rm.clamp_min(1e-30))
    raise ValueError(f"unsupported loss metric {metric!r} (mae | mse | rmse)")


def _loss_and_grad_sharded(pred, target, metric: str, n_on: int):
    """loss over [replicated on-site rows; this rank's off-site rows] of an edge-sharded crystal == the loss of the whole crystal:
    the off-site sums are added up over the ranks, the on-site rows counted once"""
    import torch.distributed as dist
    diff = pred - target
    metric = metric.lower()
    f = {"mae": torch.abs, "mse": torch.square, "rmse": torch.square}.get(metric)
    if f is None:
        raise ValueError(f"unsupported loss metric {metric!r} (mae | mse | rmse)")
    stats = torch.stack([f(diff[n_on:]).sum().double(), torch.tensor(float(diff[n_on:].numel()), dtype=torch.float64, device=diff.device)])
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    n = float(stats[1]) + diff[:n_on].numel()
    total = (stats[0] + f(diff[:n_on]).sum().double()) / n
    if metric == "mae":
        return total.to(pred.dtype), torch.sign(diff) / n
    if metric == "mse":
        return total.to(pred.dtype), 2.0 * diff / n
    rm = torch.sqrt(total)
    return rm.to(pred.dtype), diff / (n * rm.clamp_min(1e-30).to(pred.dtype))


def _sparsity_weighted(out, loss, grad):
    """the reference multiplies every hamiltonian / hamiltonian_real / hamiltonian_imag loss by predictions['sparsity_ratio'] whenever the
    head emits it (hamgnn/models/Model.py:158-162; calculate_sparsity=True is the default of the head)"""
    sr = out.get("sparsity_ratio") if hasattr(out, "get") else None
    if sr is None:
        return loss, grad
    sr = sr.to(loss.dtype)
    return loss * sr, grad * sr


@torch.no_grad()
def head_training_step(model, batch, metric: str = "mae", target: Optional[torch.Tensor] = None,
                       representation=None) -> Dict[str, torch.Tensor]:
    """One loss / gradient evaluation for the head of `model` (hamgnn_amd.models.model.Model): forward, loss(hamiltonian, target),
    backward through the head on the GPU kernels, `.grad` of every head parameter set (accumulated if already present).  The caller owns
    the optimiser: `opt.step(); opt.zero_grad()` -- the head repacks its weights on the next forward.
    representation: reuse a frozen backbone's output for this batch (it does not change while only the head is trained)."""
    head = model.output_module
    rep = representation if representation is not None else model.representation(batch)
    out = head(batch, rep)
    tgt = target if target is not None else gget(batch, "hamiltonian")
    if tgt is None:
        raise ValueError("head_training_step: the batch carries no target (Hon / Hoff or hamiltonian)")
    H = out["hamiltonian"]
    loss, gH = _loss_and_grad(H, tgt.to(H.dtype), metric)
    loss, gH = _sparsity_weighted(out, loss, gH)
    g_node, g_edge, grads = head.backward(batch, rep, gH)
    params = dict(head.named_parameters())
    for k, g in grads.items():
        p = params[k]
        g = g.reshape(p.shape).to(p.dtype)
        p.grad = g.clone() if p.grad is None else p.grad + g
    _invalidate(head)                                          # the packed weight fragments are stale once the optimiser has stepped
    return {"loss": loss, "representation": rep, "g_node_planar": g_node, "g_edge_planar_rot": g_edge}


def _sharded_sparsity_ratio(head, batch):
    """calculate_sparsity_ratio of the whole crystal from its shards: on-site rows once, off-site counts summed over the ranks"""
    import torch.distributed as dist
    z = batch.z
    n2 = head.nao_max ** 2
    src, dst = batch.edge_index
    ni = head._norb[z]
    both = head._defined[z[src]] & head._defined[z[dst]]
    off = torch.stack([torch.where(both, ni[src] * ni[dst], torch.full_like(src, n2)).sum().double(),
                       torch.tensor(float(src.numel()), dtype=torch.float64, device=z.device)])
    dist.all_reduce(off, op=dist.ReduceOp.SUM)
    eff = (ni * ni).sum().double() + off[0]
    total = (z.numel() + off[1]) * n2
    return (total / eff).to(torch.float32)


def weights_changed(model):
    """call after the optimiser stepped (training_step does it for the step that follows): the backbone's message blocks repack their
    weights on the device, everything else is dropped and repacked by the next forward"""
    rep = getattr(model, "representation", None)
    if rep is not None and hasattr(rep, "refresh_weights"):
        _invalidate(model.output_module)
        rep._pending_refresh = True                            # the optimiser steps AFTER training_step returns: refresh at the next forward
    else:
        _invalidate(model)


def allreduce_gradients(model, average: bool = True):
    """Data-parallel training (the reference's DDP, hamgnn/main.py:318-321: every rank steps the same model on its own batches): average the
    `.grad` of all parameters over the ranks of the default process group -- ONE flat bucket, i.e. one RCCL all-reduce per step (xGMI
    rings are per-link bound: few large collectives; the whole set-A model is 30 MB of fp32 gradients).  A no-op without an initialised
    process group or with one rank.  Parameters without a gradient on this rank contribute zeros."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    params = [p for p in model.parameters() if p.requires_grad]
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    o = 0
    for p in params:
        n = p.numel()
        p.grad = flat[o:o + n].reshape(p.shape).to(p.dtype)
        o += n


def _invalidate(module):
    """the packed weight fragments / adjoint tables are stale once the optimiser has stepped: repacked on the next forward"""
    from .nn import E3Linear
    for m in module.modules():
        if isinstance(m, E3Linear) and m._dp is not None and getattr(m._dp, "refreshable", False):
            m._stale = True                                    # its tables are refreshed on the device at the next use (nn.E3Linear.compile)
            continue
        for attr in ("_dp", "_dp_adj", "_adj_tabs"):             # (`_wgrad` is handed over by the blocks' compile(): its device constants are reused)
            if hasattr(m, attr):
                setattr(m, attr, None)
        if hasattr(m, "_rowprog"):                             # HamLayer's fused inference chain: its 0.1 s host build per step is not worth it while
            m._rowprog = None                                  # the weights move (the backward re-evaluates the stages one by one anyway)
            m._rowprog_off = True
        if hasattr(m, "_compiled_for"):
            m._compiled_for = None


def enable_row_programs(module):
    """after training: let the HamLayers build their fused inference chains (csrc/rowprog.hip) again -- _invalidate switches them off while the
    weights move, and the flag is sticky (validation / inference inside a training run stay on the separate kernels)"""
    for m in module.modules():
        if hasattr(m, "_rowprog_off"):
            m._rowprog_off = False
            m._rowprog = None


@torch.no_grad()
def training_step(model, batch, metric: str = "mae", target: Optional[torch.Tensor] = None, losses=None) -> Dict[str, torch.Tensor]:
    """One loss / gradient evaluation of the WHOLE model (HamGNNConvE3 backbone + non-SOC HamGNNPlusPlusOut head): forward with the
    layer inputs kept, loss(hamiltonian, target), backward through head and backbone on the GPU kernels, `.grad` of every parameter set
    (accumulated if already present).  The caller owns the optimiser (`opt.step(); opt.zero_grad()`); all packed weights are dropped
    here and repacked by the next forward (a host-side repack of every block: fine for fine-tuning runs, the thing to make
    incremental for long trainings)."""
    backbone, head = model.representation, model.output_module
    rep = backbone(batch, save_for_backward=True)
    out = head(batch, rep)
    tgt = target if target is not None else gget(batch, "hamiltonian")
    if tgt is None:
        raise ValueError("training_step: the batch carries no target (Hon / Hoff or hamiltonian)")
    H = out["hamiltonian"]
    sharded = parallel.is_sharded(batch)
    if sharded:
        # model-parallel step on an edge-sharded crystal: the on-site rows are replicated, every rank holds its own off-site rows
        if losses is not None:
            raise NotImplementedError("training_step on an edge-sharded graph: the plain hamiltonian loss")
        loss, gH = _loss_and_grad_sharded(H, tgt.to(H.dtype), metric, int(batch.z.shape[0]))
        if out.get("sparsity_ratio") is not None:              # the ratio of the WHOLE crystal (the shard's own count would differ per rank)
            sr = _sharded_sparsity_ratio(head, batch).to(loss.dtype)
            loss, gH = loss * sr, gH * sr
    elif losses is None:
        loss, gH = _sparsity_weighted(out, *_loss_and_grad(H, tgt.to(H.dtype), metric))
    else:
        # the reference's `losses` list (Model.py:150-196): [{metric, prediction, target, loss_weight}] over `hamiltonian` and / or
        # `band_energy` (the second training stage: bands of H(k) against the bands of the target Hamiltonian)
        loss, gH = H.new_zeros(()), torch.zeros_like(H)
        g_unshifted = None                                     # gradient w.r.t. the blocks before the zero-point shift (band energies)
        for spec in losses:
            w, pred = float(spec.get("loss_weight", 1.0)), spec["prediction"].lower()
            if pred == "hamiltonian":
                t_ = gget(batch, spec["target"].lower()) if spec.get("target") else tgt
                li, gi = _sparsity_weighted(out, *_loss_and_grad(H, t_.to(H.dtype), spec["metric"]))
                gH += w * gi
            elif pred in ("hamiltonian_real", "hamiltonian_imag"):
                # SOC heads: result["hamiltonian"] = [real rows; imaginary rows] (hamgnn_output.py:3621-3626 attaches the targets alike)
                if out.get(pred) is None:
                    raise ValueError(f"a {pred} loss needs a spin-orbit head")
                half = H.shape[0] // 2
                rows = slice(0, half) if pred == "hamiltonian_real" else slice(half, None)
                t_ = gget(batch, spec.get("target", pred).lower())
                li, gi = _sparsity_weighted(out, *_loss_and_grad(H[rows], t_.to(H.dtype), spec["metric"]))
                gH[rows] += w * gi
            elif pred == "band_energy":
                from . import kspace
                if out.get("band_energy") is None:
                    raise ValueError("a band_energy loss needs HamGNNPlusPlusOut(calculate_band_energy=True)")
                be = out["band_energy"]
                li, gbe = _loss_and_grad(be, gget(batch, spec.get("target", "band_energy").lower()).to(be.dtype), spec["metric"])
                edge_counts = head._global_inverse(batch)[1]
                Hb = H
                if head.zero_point_shift:
                    # the bands were computed from the blocks BEFORE the shift (hamgnn_output.py:3802-3880 precede :3971-3981) and then aligned
                    # by their mean (:3983-3985): the forward kept the unshifted rows for this re-evaluation; adjoint of the alignment = g - mean(g)
                    gbe = gbe - gbe.mean()
                    Hb = head._unshifted
                if getattr(head, "soc_switch", False):
                    # spin-orbit head: rows [real (N + E); imaginary (N + E)] of width (2 nao)^2, the bands of the stacked spinor H(k)
                    half = Hb.shape[0] // 2
                    on_r, off_r = head._split_by_crystal(batch, Hb[:half], edge_counts)
                    on_i, off_i = head._split_by_crystal(batch, Hb[half:], edge_counts)
                    gs = kspace.band_energy_backward_soc(head, on_r.contiguous(), on_i.contiguous(), off_r.contiguous(), off_i.contiguous(), batch, w * gbe)
                    gb = torch.cat([head._cat_by_crystal(batch, gs[0], gs[2], edge_counts), head._cat_by_crystal(batch, gs[1], gs[3], edge_counts)], 0)
                else:
                    on, off = head._split_by_crystal(batch, Hb, edge_counts)
                    g_on, g_off = kspace.band_energy_backward(head, on.contiguous(), off.contiguous(), batch, w * gbe)
                    gb = head._cat_by_crystal(batch, g_on, g_off, edge_counts)
                g_unshifted = gb if g_unshifted is None else g_unshifted + gb
            else:
                raise ValueError(f"training_step: losses on {pred!r} are not built (hamiltonian | hamiltonian_real | hamiltonian_imag | band_energy)")
            loss = loss + w * li
    g_node, g_edge, g_head = head.backward(batch, rep, gH, grad_unshifted=g_unshifted if losses is not None and not sharded else None)
    g_back = backbone.backward(batch, rep, g_node, g_edge)
    if sharded:                                                # per-edge parameters: sum the ranks' partial gradients (one flat bucket each)
        parallel.allreduce_edge_summed_gradients(g_head, batch)
        parallel.allreduce_edge_summed_gradients(g_back, batch)
    for mod, grads in ((head, g_head), (backbone, g_back)):
        params = dict(mod.named_parameters())
        missing = set(params) - set(grads)
        if missing:
            raise RuntimeError(f"training_step: no gradient for {sorted(missing)[:4]} ...")
        for k, g in grads.items():
            p = params[k]
            g = g.reshape(p.shape).to(p.dtype)
            p.grad = g.clone() if p.grad is None else p.grad + g
    if not sharded:                                            # (a sharded step uses the process group for the model-parallel sums)
        allreduce_gradients(model)                             # data-parallel runs: mean over ranks, one collective; else a no-op
    weights_changed(model)
    return {"loss": loss, "hamiltonian": H}

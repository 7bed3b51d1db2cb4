"""Weight gradients of a (non-lite) MessagePackBlock, first version (plan.build_message_pack_wgrad_programs).

The two materialisation programs run on the fused HIP kernels (per edge: A = cf (W x) and B = L g for every row of every super-path, in the
edge-aligned frame); everything below is reductions over the edges of products of those rows with the block's inputs -- plain GEMMs and
element-wise products on torch tensors (rocBLAS / hipBLASLt), device-agnostic so that the CPU suite runs the same code on the emulator's
output.  Gradients come back in the reference's parameter names and flat e3nn layouts:

  {node,edge}_tensor_product.weight          g_W[u, w] (path n) = c_path * sum_{e, c} x[e, u, sigma(c)] * (s cf B)[e, row(n, w), c]
  {node,edge}_linear_scaler.linear_out.weight, {node,edge}_linear_out.weight
                                             from g_L[row, w''] = sum_{e, c} (s A)[e, row, c] g[e, w'', c]  with L = Ls / sqrt(fan) @ Lo / sqrt(mul)
  {node,edge}_weight_generator.layer*.weight  last layer: h^T g_s / sqrt(H) with g_s[e, row] = sum_c A B;  hidden layers: torch.autograd on
                                             the 64-wide MLP (two dense layers per edge; the radial basis rows are inputs, not parameters)

Correct, not fast: 2 x ~80 KB of intermediates per edge, processed in chunks of 65 536 edges (10 GB), ~20 small launches per row chunk.  The fused weight-gradient kernel
 replaces this."""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import plan as P


def _block(x: torch.Tensor, off: int, mulp: int, comps: Sequence[int], nch: int) -> torch.Tensor:
    """planar rows -> [E, len(comps), nch]: channels 0..nch of the given components of the irrep block at `off`.  The components of a
    block are `mulp` floats apart: a contiguous ascending run of components is a VIEW (no launch), a descending one a flipped view."""
    comps = list(comps)
    lo, hi = min(comps), max(comps)
    if comps == list(range(lo, hi + 1)) or comps == list(range(hi, lo - 1, -1)):
        v = x[:, off + lo * mulp:off + (hi + 1) * mulp].unflatten(1, (hi - lo + 1, mulp))[:, :, :nch]
        return v if comps[0] == lo else v.flip(1)
    return torch.stack([x[:, off + a * mulp:off + a * mulp + nch] for a in comps], 1)


def radial_mlp(rbf: torch.Tensor, layers: List[torch.Tensor], act_cst: float) -> torch.Tensor:
    """FullyConnectedNet hidden layers as torch ops (e3nn: x @ W / sqrt(fan_in), normalised SiLU): the differentiable twin of
    hg_radial_hidden; `layers` are the RAW parameters [h_in, h_out]"""
    h = rbf
    for W in layers:
        h = torch.nn.functional.silu(h @ (W / math.sqrt(W.shape[0]))) * act_cst
    return h


def _chunk_consts(c, device, dtype):
    """per row chunk, once per (TPWeightGrad, device): the index / coefficient tensors of its rows"""
    key = (str(device), dtype)
    d = c.get("_dev")
    if d is not None and d[0] == key:
        return d[1]
    sp, r0, r1 = c["sp"], c["r0"], c["r1"]
    meta = sp["meta"][r0:r1]
    ch = sp["ch"][r0:r1]
    mk, mi2 = sp["mk"], c["nsrc"] * sp["mi"]
    # flat TP weight of row r (path pn, mid channel w), input channel u: woff[pn] + u mk + w;  row of the k-block of L: lrow
    base = np.array([sp["woff"][m[0]] + m[1] for m in meta], dtype=np.int64)
    tp_idx = base[:, None] + np.arange(mi2, dtype=np.int64)[None, :] * mk
    out = dict(ch=torch.as_tensor(ch, device=device), cf=torch.as_tensor(sp["cf"][r0:r1].T.copy(), device=device, dtype=dtype),
               ch_slice=slice(int(ch[0]), int(ch[-1]) + 1) if np.array_equal(ch, np.arange(ch[0], ch[0] + len(ch))) else None,
               tp_idx=torch.as_tensor(tp_idx.reshape(-1), device=device),
               cpath=torch.as_tensor([m[2] for m in meta], device=device, dtype=dtype)[:, None],
               lrow=torch.as_tensor([m[3] for m in meta], device=device, dtype=torch.int64))
    c["_dev"] = (key, out)
    return out


def weight_grads_from_rows(chunks, Arows: torch.Tensor, Brows: torch.Tensor, srcs: Sequence[torch.Tensor], g: torch.Tensor,
                           S: Dict[str, torch.Tensor], acc: Dict[str, torch.Tensor], gs_all: Dict[str, torch.Tensor], irreps_out, gx=None):
    """One chunk of edges.  chunks: plan.build_tp_wgrad_programs' bookkeeping; Arows / Brows: the two materialised row tensors
    [E, out_dim]; srcs = the program's source rows by slot (message pack: x_sender, x_receiver, f), planar, edge frame; g: gradient rows
    (edge frame, planar(irreps_out)); S[branch] = h @ W3 / sqrt(H) [E, n_channels]; acc: running sums of the per-path / per-k gradients
    (updated in place); gs_all[branch]: [E, n_channels] (filled); gx: optional list of zero tensors like srcs -- the gradient with
    respect to the source rows is accumulated there (W^T (s cf B); used for the 0e-only embedding input, where it is a few columns)."""
    gl = P.PlanarLayout(irreps_out)
    for c in chunks:
        sp, r0, r1 = c["sp"], c["r0"], c["r1"]
        n, mm, li, lk, mk, k, i = r1 - r0, sp["mm"], sp["li"], sp["lk"], sp["mk"], sp["k"], sp["i"]
        nc = 2 * mm + 1
        cols = [lk - mm + cc for cc in range(nc)]
        K = _chunk_consts(c, Arows.device, Arows.dtype)
        A = _block(Arows, c["out_off"], c["out_mulp"], cols, n)                    # [E, nc, n]
        B = _block(Brows, c["out_off"], c["out_mulp"], cols, n)
        chs = K["ch_slice"] if K["ch_slice"] is not None else K["ch"]
        s = S[c["branch"]][:, chs]                                                 # [E, n]
        gs_all[c["branch"]][:, chs] = (A * B).sum(1)
        Gk = _block(g, gl.off[k], gl.mulp[k], cols, mk)                            # [E, nc, mk]
        gL = torch.einsum("ecn,ecw->nw", A * s[:, None, :], Gk)                    # rows x mul_k
        T1 = B * s[:, None, :] * K["cf"][None]
        lay = c["lay"]
        comps = [(li + mm - cc) if sp["par"] else (li - mm + cc) for cc in range(nc)]
        X = torch.cat([_block(srcs[sl], lay.off[i], lay.mulp[i], comps, sp["mi"]) for sl in c["srcs"]], 2)    # [E, nc, nsrc * mi]
        gW = torch.einsum("ecn,ecu->nu", T1, X)                                    # rows x (nsrc mul_i)
        if gx is not None:
            Wr = torch.as_tensor(sp["W"][r0:r1], device=T1.device, dtype=T1.dtype)   # [n, nsrc * mi], path normalisation included
            GX = torch.einsum("ecn,nu->ecu", T1, Wr)
            for q, sl in enumerate(c["srcs"]):
                for cc, a in enumerate(comps):
                    o = lay.off[i] + a * lay.mulp[i]
                    gx[sl][:, o:o + sp["mi"]] += GX[:, cc, q * sp["mi"]:(q + 1) * sp["mi"]]
        name = c["branch"]
        acc[f"{name}_tp"].index_add_(0, K["tp_idx"], (gW * K["cpath"]).reshape(-1))
        acc[f"{name}_L"][k].index_add_(0, K["lrow"], gL)


class TPWeightGrad:
    """holds the two materialisation programs of the weighted tensor-product branches of one block and turns (inputs, output gradient)
    into parameter gradients.  branches: plan.message_pack_wgrad_branches / plan.embedding_wgrad_branches."""

    def __init__(self, sd: Dict[str, np.ndarray], branches, irreps_sh, irreps_out):
        self.sd = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
        self.irreps_out = P.Irreps(irreps_out)
        self.branches = branches
        self.H = int(branches[0]["w3"].shape[0])
        self.irreps_sh = P.Irreps(irreps_sh)
        self.progA, self.progB, self.chunks = P.build_tp_wgrad_programs(branches, irreps_sh, irreps_out, self.H)

    def adopt_constants(self, old: "TPWeightGrad"):
        """after a weight update: the per-chunk index / coefficient tensors already on the device depend on the irreps only -- take them
        over from the previous instance instead of re-uploading ~2 000 small arrays per block and step"""
        if old is None or len(old.chunks) != len(self.chunks):
            return self
        for c, o in zip(self.chunks, old.chunks):
            if "_dev" in o and (c["branch"], c["r0"], c["r1"], c["sp"]["i"], c["sp"]["k"]) == (o["branch"], o["r0"], o["r1"], o["sp"]["i"], o["sp"]["k"]):
                c["_dev"] = o["_dev"]
        return self

    def new_acc(self, device, dtype):
        acc = {}
        for b in self.branches:
            name = b["name"]
            acc[f"{name}_tp"] = torch.zeros(self.sd[b["keys"]["tp"]].size, device=device, dtype=dtype)      # the reference's flat layout
            acc[f"{name}_L"] = {}
            for c in self.chunks:
                if c["branch"] == name and c["sp"]["k"] not in acc[f"{name}_L"]:
                    off, fan = c["sp"]["lin"]
                    acc[f"{name}_L"][c["sp"]["k"]] = torch.zeros(fan, c["sp"]["mk"], device=device, dtype=dtype)
        return acc

    def finish(self, acc, gW3: Dict[str, Dict[str, torch.Tensor]], dev, dt) -> Dict[str, torch.Tensor]:
        """per-path / per-k sums -> the reference's flat parameter gradients"""
        out = {}
        for b in self.branches:
            name, keys = b["name"], b["keys"]
            out[keys["tp"]] = acc[f"{name}_tp"]
            Ls_flat = torch.as_tensor(self.sd[keys["ls"]], device=dev, dtype=dt)
            gLs = torch.zeros_like(Ls_flat)
            if keys["lo"] is not None:
                Lo_flat = torch.as_tensor(self.sd[keys["lo"]], device=dev, dtype=dt)
                gLo = torch.zeros_like(Lo_flat)
            seen = set()
            for c in self.chunks:
                sp = c["sp"]
                if c["branch"] != name or sp["k"] in seen:
                    continue
                seen.add(sp["k"])
                (off, fan), lo_off, mk = sp["lin"], sp["lo_off"], sp["mk"]
                gL = acc[f"{name}_L"][sp["k"]]                 # d / d (Ls / sqrt(fan) [@ Lo / sqrt(mk)])
                if keys["lo"] is None:
                    gLs[off:off + fan * mk] = (gL / math.sqrt(fan)).reshape(-1)
                    continue
                Ls = Ls_flat[off:off + fan * mk].reshape(fan, mk) / math.sqrt(fan)
                Lo = Lo_flat[lo_off:lo_off + mk * mk].reshape(mk, mk) / math.sqrt(mk)
                gLs[off:off + fan * mk] = ((gL @ Lo.t()) / math.sqrt(fan)).reshape(-1)
                gLo[lo_off:lo_off + mk * mk] = ((Ls.t() @ gL) / math.sqrt(mk)).reshape(-1)
            out[keys["ls"]] = gLs
            if keys["lo"] is not None:
                out[keys["lo"]] = gLo
            out.update(gW3[name])
        return out


class MessagePackWeightGrad(TPWeightGrad):
    def __init__(self, sd: Dict[str, np.ndarray], irreps_node, irreps_edge, irreps_sh, irreps_out):
        super().__init__(sd, P.message_pack_wgrad_branches(sd, irreps_node, irreps_edge), irreps_sh, irreps_out)


def tp_weight_grads(wg: TPWeightGrad, run_program, srcs: Sequence[torch.Tensor], g, rbf, act_cst: float, chunk: int = 65536, want_gx: bool = False):
    """run_program(prog, sources, h_node, h_edge) -> rows [E_chunk, out_dim] (the HIP kernels on the GPU, the emulator in the CPU suite).
    srcs: planar edge-frame input rows of the block by source slot; g: gradient of its output rows (edge frame); rbf: radial basis rows.
    Returns the parameter gradients ({reference name: flat gradient}) -- and, with want_gx, the gradients of the source rows."""
    dev, dt = g.device, g.dtype
    E = g.shape[0]
    sd, H = wg.sd, wg.H
    acc = wg.new_acc(dev, dt)
    gen, gkeys = {}, {}
    for b in wg.branches:
        pre = b["keys"]["gen"]
        gkeys[b["name"]] = sorted(k for k in sd if k.startswith(pre + ".layer") and k.endswith(".weight"))
        gen[b["name"]] = [torch.as_tensor(sd[k], device=dev, dtype=dt).requires_grad_() for k in gkeys[b["name"]]]
    gW3 = {name: {} for name in gen}
    gh_hidden = {name: torch.zeros(E, H, device=dev, dtype=dt) for name in gen}
    gW3_last = {name: torch.zeros_like(gen[name][-1]) for name in gen}
    gx = [torch.zeros_like(t) for t in srcs] if want_gx else None
    for e0 in range(0, E, chunk):
        sl = slice(e0, min(E, e0 + chunk))
        n = sl.stop - sl.start
        with torch.no_grad():
            h = {name: radial_mlp(rbf[sl], [w.detach() for w in gen[name][:-1]], act_cst) for name in gen}
            S = {name: h[name] @ (gen[name][-1].detach() / math.sqrt(H)) for name in gen}
            ones = torch.zeros(n, wg.progA.hidden_pad, device=dev, dtype=dt)
            ones[:, 0] = 1.0
            part = [t[sl] for t in srcs]
            Arows = run_program(wg.progA, part, ones, ones)
            Brows = run_program(wg.progB, [g[sl]], ones, ones)
            gs_all = {name: torch.zeros(n, gen[name][-1].shape[1], device=dev, dtype=dt) for name in gen}
            weight_grads_from_rows(wg.chunks, Arows, Brows, part, g[sl], S, acc, gs_all, wg.irreps_out,
                                   gx=None if gx is None else [t[sl] for t in gx])
            for name in gen:
                gW3_last[name] += h[name].t() @ gs_all[name] / math.sqrt(H)
                gh_hidden[name][sl] = gs_all[name] @ (gen[name][-1].detach().t() / math.sqrt(H))
    for name in gen:                                           # hidden layers of the radial MLPs: two dense layers per edge, torch.autograd
        hidden = gen[name][:-1]
        ks = gkeys[name]
        if hidden:
            with torch.enable_grad():                          # callers run under no_grad (training_step)
                hfull = radial_mlp(rbf, hidden, act_cst)
                grads = torch.autograd.grad(hfull, hidden, grad_outputs=gh_hidden[name], allow_unused=True)
            for k, w, gk in zip(ks[:-1], hidden, grads):
                gW3[name][k] = gk if gk is not None else torch.zeros_like(w)
        gW3[name][ks[-1]] = gW3_last[name]
    out = wg.finish(acc, gW3, dev, dt)
    return (out, gx) if want_gx else out


def block_weight_grads(wg: MessagePackWeightGrad, run_program, xs, xd, f, g, rbf, act_cst: float, chunk: int = 65536) -> Dict[str, torch.Tensor]:
    """MessagePackBlock: sources = (sender rows, receiver rows, edge rows), all planar in the edge frame"""
    return tp_weight_grads(wg, run_program, [xs, xd, f], g, rbf, act_cst, chunk)

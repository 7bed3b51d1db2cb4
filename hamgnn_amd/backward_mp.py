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
from . import ops
from .ops import scatter_rows as _scatter_rows


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


def _build_groups(wg, device, dtype):
    """Row chunks of like shape are processed TOGETHER (one gather / product / einsum / index_add per group instead of per chunk: the
    per-chunk version was launch-bound, ~20 small launches x 340 chunks per block).  Group key = (branch, columns, parity);
    inside a group the rows n, the output multiplicity mk and the input channels U are padded to the group maximum.
    Padding costs no masks: padded rows read the spare ZERO column of S (scale 0), padded outputs land in the spare slot of the
    accumulators.  Everything here depends on the irreps only, not on the weights."""
    gl = P.PlanarLayout(wg.irreps_out)
    slot_base = {}                                             # column offset of each source slot in the concatenated source rows
    groups = {}
    for j, c in enumerate(wg.chunks):
        sp = c["sp"]
        n, U = c["r1"] - c["r0"], c["nsrc"] * sp["mi"]
        groups.setdefault((c["branch"], 2 * sp["mm"] + 1, sp["par"]), []).append(j)
    out = []
    for (branch, nc, par), js in groups.items():
        b = next(b for b in wg.branches if b["name"] == branch)
        lay = b["lay"]
        nch = int(np.asarray(b["w3"]).shape[1])
        tp_size = int(wg.sd[b["keys"]["tp"]].size) if b["keys"]["tp"] is not None else 0     # uvu branches carry no tensor-product weights
        ls_size = int(wg.sd[b["keys"]["ls"]].size)
        # bound the gathered tensors to ~48 KB per edge
        cs = [wg.chunks[j] for j in js]
        idx_of = {id(c): j for j, c in zip(js, cs)}
        per = max(nc * max(max(c["r1"] - c["r0"] for c in cs), max(c["nsrc"] * c["sp"]["mi"] for c in cs)), 1)
        gmax = max(1, 12288 // per)
        for g0 in range(0, len(cs), gmax):
            part = cs[g0:g0 + gmax]
            G = len(part)
            N = max(c["r1"] - c["r0"] for c in part)
            MK = max(c["sp"]["mk"] for c in part)
            U = max(c["nsrc"] * c["sp"]["mi"] for c in part)
            a_idx = np.zeros((G, N, nc), np.int64)             # FEATURE-major: the reductions run on transposed rows ([feature, edge]),
            g_idx = np.zeros((G, MK, nc), np.int64)            # so a gather lands directly in the [G, rows, (column, edge)] operand layout
            x_idx = np.zeros((G, U, nc), np.int64)             # of the batched GEMMs (no permute / contiguous copies)
            ch_idx = np.full((G, N), nch, np.int64)            # spare row of S^T / gs_all^T
            cf = np.zeros((G, N, nc))
            cpath = np.zeros((G, N, 1))
            tp_idx = np.full((G, N, U), tp_size, np.int64)     # spare slot of the flat TP-weight accumulator
            l_idx = np.full((G, N, MK), ls_size, np.int64)     # spare slot of the flat L' accumulator
            for q, c in enumerate(part):
                sp, r0, r1 = c["sp"], c["r0"], c["r1"]
                n, mm, li, lk, mk, k, i, mi = r1 - r0, sp["mm"], sp["li"], sp["lk"], sp["mk"], sp["k"], sp["i"], sp["mi"]
                u = c["nsrc"] * mi
                cols = np.array([lk - mm + cc for cc in range(nc)])
                comps = np.array([(li + mm - cc) if par else (li - mm + cc) for cc in range(nc)])
                a_idx[q, :n, :] = c["out_off"] + cols[None, :] * c["out_mulp"] + np.arange(n)[:, None]
                g_idx[q, :mk, :] = gl.off[k] + cols[None, :] * gl.mulp[k] + np.arange(mk)[:, None]
                for t, sl in enumerate(c["srcs"]):
                    base = slot_base.setdefault((branch, sl), len([1 for key in slot_base if key[0] == branch]) * lay.dim)
                    x_idx[q, t * mi:(t + 1) * mi, :] = base + lay.off[i] + comps[None, :] * lay.mulp[i] + np.arange(mi)[:, None]
                ch_idx[q, :n] = sp["ch"][r0:r1]
                cf[q, :n, :] = sp["cf"][r0:r1]
                meta = sp["meta"][r0:r1]
                cpath[q, :n, 0] = [m[2] for m in meta]
                if b["keys"]["tp"] is not None:
                    base_tp = np.array([sp["woff"][m[0]] + m[1] for m in meta], dtype=np.int64)
                    tp_idx[q, :n, :u] = base_tp[:, None] + np.arange(u, dtype=np.int64)[None, :] * mk
                off, fan = sp["lin"]
                lrow = np.array([m[3] for m in meta], dtype=np.int64)
                l_idx[q, :n, :mk] = off + lrow[:, None] * mk + np.arange(mk, dtype=np.int64)[None, :]
            t_ = lambda arr, dt=None: torch.as_tensor(arr, device=device, dtype=dt)
            out.append(dict(branch=branch, srcs=sorted({sl for c in part for sl in c["srcs"]}, key=lambda sl: slot_base[(branch, sl)]),
                            a_idx=t_(a_idx), g_idx=t_(g_idx), x_idx=t_(x_idx), ch_idx=t_(ch_idx), cf=t_(cf[..., None], dtype), cpath=t_(cpath, dtype),
                            tp_idx=t_(tp_idx.reshape(-1)), l_idx=t_(l_idx.reshape(-1)), cidx=[idx_of[id(c)] for c in part], shape=(G, N, U)))
    return out


def weight_grads_from_rows(wg, ArowsT: torch.Tensor, BrowsT: torch.Tensor, srcsT: Sequence[torch.Tensor], gT: torch.Tensor,
                           ST: Dict[str, torch.Tensor], acc: Dict[str, torch.Tensor], gs_allT: Dict[str, torch.Tensor], gxT=None):
    """One chunk of edges, everything FEATURE-major ([feature, edge]: the transposes are made once per chunk by the caller).
    ArowsT / BrowsT: the two materialised row tensors [out_dim, E]; srcsT = the program's source rows by slot (message pack: x_sender,
    x_receiver, f), planar, edge frame; gT: gradient rows (edge frame, planar(irreps_out)); ST[branch] = [h @ W3 / sqrt(H); 0]
    [n_channels + 1, E]; acc: flat running sums of the TP-weight and L' gradients (+ one spare slot each, updated in place);
    gs_allT[branch]: [n_channels + 1, E] (filled); gxT: optional list of zero tensors like srcsT -- the gradient with respect to the
    source rows is accumulated there (W^T (s cf B); used for the 0e-only embedding input, where it is a few rows)."""
    if getattr(wg, "_groups", None) is None or wg._groups[0] != (str(ArowsT.device), ArowsT.dtype):
        wg._groups = ((str(ArowsT.device), ArowsT.dtype), _build_groups(wg, ArowsT.device, ArowsT.dtype))
    E = ArowsT.shape[1]
    xcat = {}
    for grp in wg._groups[1]:
        name = grp["branch"]
        if name not in xcat:
            slots = sorted({sl for g_ in wg._groups[1] if g_["branch"] == name for sl in g_["srcs"]})
            xcat[name] = (srcsT[slots[0]] if len(slots) == 1 else torch.cat([srcsT[sl] for sl in slots], 0), slots)
        G, N, nc = grp["a_idx"].shape
        X = xcat[name][0][grp["x_idx"]]                                            # [G, U, nc, E]
        A, B = ArowsT[grp["a_idx"]], BrowsT[grp["a_idx"]]                          # [G, N, nc, E]
        s = ST[name][grp["ch_idx"]][:, :, None, :]                                 # [G, N, 1, E]  (0 on padded rows)
        gs_allT[name][grp["ch_idx"].reshape(-1)] = (A * B).sum(2).reshape(G * N, E)
        Gk = gT[grp["g_idx"]]                                                      # [G, MK, nc, E]
        gL = torch.bmm((A * s).reshape(G, N, nc * E), Gk.reshape(G, -1, nc * E).transpose(1, 2))          # [G, N, MK]
        T1 = (B * s * grp["cf"]).reshape(G, N, nc * E)
        gW = torch.bmm(T1, X.reshape(G, -1, nc * E).transpose(1, 2))                                        # [G, N, U]
        acc[f"{name}_tp"] += _scatter_rows(grp["tp_idx"], (gW * grp["cpath"]).reshape(-1), acc[f"{name}_tp"].shape[0])      # fixed order
        acc[f"{name}_L"] += _scatter_rows(grp["l_idx"], gL.reshape(-1), acc[f"{name}_L"].shape[0])
        if gxT is not None:
            Wg_np = np.zeros(grp["shape"])                                         # [G, N, U] from the CURRENT weights, path normalisation included
            for q, j in enumerate(grp["cidx"]):
                c = wg.chunks[j]
                Wr = c["sp"]["W"][c["r0"]:c["r1"]]
                Wg_np[q, :Wr.shape[0], :Wr.shape[1]] = Wr
            Wg = torch.as_tensor(Wg_np, device=T1.device, dtype=T1.dtype)
            GX = torch.bmm(Wg.transpose(1, 2), T1).reshape(-1, E)                  # [G U nc, E]
            rows = grp["x_idx"].reshape(-1)
            lay_dim = xcat[name][0].shape[0] // len(xcat[name][1])
            for t, sl in enumerate(xcat[name][1]):                                 # rows of slot t of the concatenated sources
                sel = (rows >= t * lay_dim) & (rows < (t + 1) * lay_dim)
                gxT[sl] += _scatter_rows(rows[sel] - t * lay_dim, GX[sel], gxT[sl].shape[0], persistent=False)      # fixed summation order (no float atomics)


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

    params_dev = None                                          # optional {name: device tensor}: the CURRENT parameters (training: no host round trip)

    def param(self, key, device, dtype):
        if self.params_dev is not None and key in self.params_dev:
            return self.params_dev[key].detach().to(device=device, dtype=dtype)
        return torch.as_tensor(self.sd[key], device=device, dtype=dtype)

    def adopt_constants(self, old: "TPWeightGrad"):
        """after a weight update: the grouped index / coefficient tensors already on the device depend on the irreps only -- take them
        over from the previous instance instead of rebuilding and re-uploading them per block and step"""
        sig = lambda w: [(c["branch"], c["r0"], c["r1"], c["sp"]["i"], c["sp"]["k"]) for c in w.chunks]
        if old is not None and getattr(old, "_groups", None) is not None and sig(old) == sig(self):
            self._groups = old._groups
            self._groups_adopted = True
        return self

    def new_acc(self, device, dtype):
        """flat accumulators in the reference's layouts (+ one spare slot that padded group entries write to)"""
        acc = {}
        for b in self.branches:
            acc[f"{b['name']}_tp"] = torch.zeros((self.sd[b["keys"]["tp"]].size if b["keys"]["tp"] is not None else 0) + 1, device=device, dtype=dtype)
            acc[f"{b['name']}_L"] = torch.zeros(self.sd[b["keys"]["ls"]].size + 1, device=device, dtype=dtype)     # d / d L', linear_scaler's layout
        return acc

    def finish(self, acc, gW3: Dict[str, Dict[str, torch.Tensor]], dev, dt) -> Dict[str, torch.Tensor]:
        """per-path / per-k sums -> the reference's flat parameter gradients"""
        out = {}
        for b in self.branches:
            name, keys = b["name"], b["keys"]
            if keys["tp"] is not None:
                out[keys["tp"]] = acc[f"{name}_tp"][:-1]
            Ls_flat = self.param(keys["ls"], dev, dt).reshape(-1)
            gLs = torch.zeros_like(Ls_flat)
            if keys["lo"] is not None:
                Lo_flat = self.param(keys["lo"], dev, dt).reshape(-1)
                gLo = torch.zeros_like(Lo_flat)
            if keys["lo"] is not None and dt == torch.float32 and ops.use_block_gemm(Ls_flat):
                # the two products per output irrep as TWO launches of csrc/block_gemm.hip per branch (26 library GEMMs + their slicing before)
                cache = self.__dict__.setdefault("_bg_finish", {})
                if (name, dev) not in cache:
                    ua, ub, seen_ = [], [], set()
                    for c in self.chunks:
                        sp = c["sp"]
                        if c["branch"] != name or sp["k"] in seen_:
                            continue
                        seen_.add(sp["k"])
                        (off, fan), lo_off, mk = sp["lin"], sp["lo_off"], sp["mk"]
                        sc = 1.0 / (math.sqrt(fan) * math.sqrt(mk))
                        ua.append((off, mk, 0, lo_off, mk, 1, off, mk, fan, mk, mk, sc))          # gLs_k = gL_k @ Lo_k^T
                        ub.append((off, mk, 1, off, mk, 0, lo_off, mk, mk, mk, fan, sc))          # gLo_k = Ls_k^T @ gL_k
                    cache[(name, dev)] = (ops.BlockGemm(ua, dev), ops.BlockGemm(ub, dev))
                bga, bgb = cache[(name, dev)]
                gL_flat = acc[f"{name}_L"].contiguous()
                ops.block_gemm(bga, gL_flat, Lo_flat.contiguous(), gLs)
                ops.block_gemm(bgb, Ls_flat.contiguous(), gL_flat, gLo)
                out[keys["ls"]] = gLs
                out[keys["lo"]] = gLo
                out.update(gW3[name])
                continue
            seen = set()
            for c in self.chunks:
                sp = c["sp"]
                if c["branch"] != name or sp["k"] in seen:
                    continue
                seen.add(sp["k"])
                (off, fan), lo_off, mk = sp["lin"], sp["lo_off"], sp["mk"]
                gL = acc[f"{name}_L"][off:off + fan * mk].reshape(fan, mk)      # d / d (Ls / sqrt(fan) [@ Lo / sqrt(mk)])
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
        gen[b["name"]] = [wg.param(k, dev, dt).clone().requires_grad_() for k in gkeys[b["name"]]]
    gW3 = {name: {} for name in gen}
    gh_hidden = {name: torch.zeros(E, H, device=dev, dtype=dt) for name in gen}
    gW3_last = {name: torch.zeros_like(gen[name][-1]) for name in gen}
    gx = [torch.zeros_like(t) for t in srcs] if want_gx else None
    for e0 in range(0, E, chunk):
        sl = slice(e0, min(E, e0 + chunk))
        n = sl.stop - sl.start
        with torch.no_grad():
            h = {name: radial_mlp(rbf[sl], [w.detach() for w in gen[name][:-1]], act_cst) for name in gen}
            S = {name: torch.nn.functional.pad(h[name] @ (gen[name][-1].detach() / math.sqrt(H)), (0, 1)) for name in gen}   # + the zero column
            ST = {name: v.t().contiguous() for name, v in S.items()}
            ones = torch.zeros(n, wg.progA.hidden_pad, device=dev, dtype=dt)
            ones[:, 0] = 1.0
            part = [t[sl] for t in srcs]
            Arows = run_program(wg.progA, part, ones, ones)
            Brows = run_program(wg.progB, [g[sl]], ones, ones)
            gs_allT = {name: torch.zeros(gen[name][-1].shape[1] + 1, n, device=dev, dtype=dt) for name in gen}
            gxT = None if gx is None else [torch.zeros(t.shape[1], n, device=dev, dtype=dt) for t in gx]
            tr = lambda t: t.t().contiguous()                  # feature-major copies, once per chunk of edges
            weight_grads_from_rows(wg, tr(Arows), tr(Brows), [tr(t) for t in part], tr(g[sl]), ST, acc, gs_allT, gxT=gxT)
            del Arows, Brows
            gs_all = {name: v[:-1].t() for name, v in gs_allT.items()}
            if gx is not None:
                for t, tT in zip(gx, gxT):
                    t[sl] += tT.t()
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


def _ht_times(h: torch.Tensor, gs: torch.Tensor) -> torch.Tensor:
    """h^T gs ([H, E] @ [E, C]) for the last radial layer's weight gradient.  With E = 44 k and H = 64 the library's single GEMM has nothing to parallelise over but the long K:
    16 batched partial products + a fixed-order sum run at 2.9 instead of 1.8 TB/s of gs (profiles/r06_training.md); deterministic (no split-K atomics)."""
    E = h.shape[0]
    if E < 4096:
        return h.t() @ gs
    k = 16
    Ek = E // k * k
    out = torch.bmm(h[:Ek].reshape(k, Ek // k, h.shape[1]).transpose(1, 2), gs[:Ek].reshape(k, Ek // k, gs.shape[1])).sum(0)
    if Ek < E:
        out = out + h[Ek:].t() @ gs[Ek:]
    return out


def tp_weight_grads_fused(wg: TPWeightGrad, wf, run_wgrad, srcs: Sequence[torch.Tensor], g, rbf, act_cst: float, chunk: int = 1 << 20, hidden=None):
    """the same gradients as tp_weight_grads through the FUSED kernel (csrc/tp_wgrad.hip, plan.WgFused `wf`): nothing per edge is
    materialised except gs (the gradient with respect to the last radial layer's output).  run_wgrad(srcs_by_slot, g, h_node, h_edge) ->
    (acc [splits, acc_floats], [gs per branch]) -- ops.tp_wgrad on the GPU, the numpy twin (tests/emu.py:run_wgrad_fused) in the CPU suite.
    wf: the plan object or ops.DeviceWgFused (same attribute names; its gather maps are device tensors: no upload per step).
    srcs: planar edge-frame source rows by slot; hidden: optional {branch: hidden rows [E, H]} (else recomputed from rbf)."""
    dev, dt = g.device, g.dtype
    E = g.shape[0]
    sd, H = wg.sd, wg.H
    gen, gkeys = {}, {}
    for b in wg.branches:
        pre = b["keys"]["gen"]
        gkeys[b["name"]] = sorted(k for k in sd if k.startswith(pre + ".layer") and k.endswith(".weight"))
        gen[b["name"]] = [wg.param(k, dev, dt).clone().requires_grad_() for k in gkeys[b["name"]]]
    names = [b["name"] for b in wg.branches]
    plan_wf = getattr(wf, "wf", wf)
    alive = getattr(wf, "_alive_idx", None)                     # per branch: None (all channels) or the written channels as an index tensor
    if alive is None or alive[0] != str(dev):
        rngs = getattr(plan_wf, "ch_ranges", None)
        idxs = []
        for bi in range(len(wg.branches)):
            n_alive = sum(b_ - a_ for a_, b_ in rngs[bi]) if rngs is not None else plan_wf.nch[bi]
            idxs.append(None if n_alive > 0.9 * plan_wf.nch[bi] else torch.tensor([c for a_, b_ in rngs[bi] for c in range(a_, b_)], dtype=torch.long, device=dev))
        alive = (str(dev), idxs)
        try:
            wf._alive_idx = alive
        except AttributeError:
            pass
    alive = alive[1]
    gW3 = {name: {} for name in gen}
    gh_hidden = {name: torch.zeros(E, H, device=dev, dtype=dt) for name in gen}
    gW3_last = {name: torch.zeros_like(gen[name][-1]) for name in gen}
    flat = None
    for e0 in range(0, E, chunk):
        sl = slice(e0, min(E, e0 + chunk))
        with torch.no_grad():
            h = {name: (hidden[name][sl, :H] if hidden is not None else radial_mlp(rbf[sl], [w.detach() for w in gen[name][:-1]], act_cst)) for name in gen}
            by_mlp = {b["mlp"]: h[b["name"]] for b in wg.branches}
            acc, gs = run_wgrad([t[sl] if t is not None else None for t in srcs], g[sl], by_mlp[0], by_mlp.get(1))
            part = acc.sum(0)
            flat = part if flat is None else flat + part
            for bi, name in enumerate(names):
                # the last radial layer: g_W3 = h^T gs, g_h = gs W3^T -- over the radial channels some row tile wrote (tables built without the super-paths
                # of structurally zero inputs leave most columns of gs at zero: plan.build_tp_wgrad_fused zero_inputs; first ConvBlock of set-A: 152 of 3 589)
                W3 = gen[name][-1].detach()
                idx = alive[bi]
                if idx is None:
                    gW3_last[name] += _ht_times(h[name], gs[bi]) / math.sqrt(H)
                    gh_hidden[name][sl] = gs[bi] @ (W3.t() / math.sqrt(H))
                else:
                    gsc = gs[bi].index_select(1, idx)
                    gW3_last[name].index_add_(1, idx, _ht_times(h[name], gsc) / math.sqrt(H))     # (distinct columns: no two terms meet, the order is fixed)
                    gh_hidden[name][sl] = gsc @ (W3.index_select(1, idx).t() / math.sqrt(H))
    for name in gen:                                           # hidden layers of the radial MLPs: two dense layers per edge, torch.autograd
        hid, ks = gen[name][:-1], gkeys[name]
        if hid:
            with torch.enable_grad():
                hfull = radial_mlp(rbf, hid, act_cst)
                grads = torch.autograd.grad(hfull, hid, grad_outputs=gh_hidden[name], allow_unused=True)
            for k, w, gk in zip(ks[:-1], hid, grads):
                gW3[name][k] = gk if gk is not None else torch.zeros_like(w)
        gW3[name][ks[-1]] = gW3_last[name]
    acc_out = {}
    as_t = lambda a, d: a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a), device=dev, dtype=d)
    for bi, name in enumerate(names):                          # fixed-order gather of the (<= 4) edge-tile copies of every parameter's slot
        tp = flat[as_t(wf.tp_pos[bi], torch.long)].sum(1) * as_t(wf.tp_scale[bi], dt)
        acc_out[f"{name}_tp"] = torch.cat([tp, tp.new_zeros(1)])
        acc_out[f"{name}_L"] = torch.cat([flat[as_t(wf.l_pos[bi], torch.long)].sum(1), flat.new_zeros(1)])
    return wg.finish(acc_out, gW3, dev, dt)


def block_weight_grads(wg: MessagePackWeightGrad, run_program, xs, xd, f, g, rbf, act_cst: float, chunk: int = 65536) -> Dict[str, torch.Tensor]:
    """MessagePackBlock: sources = (sender rows, receiver rows, edge rows), all planar in the edge frame"""
    return tp_weight_grads(wg, run_program, [xs, xd, f], g, rbf, act_cst, chunk)

"""Tables of the fused weight-gradient kernel (csrc/tp_wgrad.hip) and the materialisation programs of the fallback route."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ..so3 import Irreps
from ._blas import single_thread_blas
from .layout import IT_TP, PlanarLayout, ceil_div, rtm_max
from .program import _add_item, _cf_block, _frag_A, _tp_superpaths, new_program, seg_rows_cap, use_x4
from .message_pack import SRC_F, SRC_XD, SRC_XS, _last_layer

@single_thread_blas
def build_tp_wgrad_programs(branches, irreps_sh, irreps_out, H: int):
    """WEIGHT gradients of the weighted tensor-product branches of a (non-lite) MessagePackBlock / the embedding TP, first version
    (SURVEY 8f-3): two programs for the existing fused kernels that MATERIALISE, per edge, what the reference's unfused graph holds anyway --
      program A (sources: the branch inputs, edge frame):                      A[row, c] = cf[row, c] (W x)[row, c]   (radial scale 1, L' = 1)
      program B (source: the gradient of the block's output rows, edge frame): B[row, c] = (L g)[row, c]              (cf = 1, scale 1)
    for every row (= (e3nn path, mid channel)) of every super-path, both in the SAME output layout (one output "irrep" (rows, l_k, p_k)
    per row chunk, columns centred like the forward's tiles).  The gradients are then reductions over the edges of products of these
    rows with the inputs (plain library GEMMs, hamgnn_amd/backward_mp.py):
      g_s[row] = sum_c A B,   g_L = (s A)^T g,   g_W = x^T (s cf B),   g_W3 = h^T g_s,   g_h = g_s W3^T.
    Correct, not fast (140 KB of intermediates per edge and branch): the fused weight-gradient kernel is the next step.
    branches: dicts {name, nsrc, srcs, lay, mlp, tp_w, w3 (raw last radial layer), ls_w, lo_w | None}.
    Returns (program A, program B, chunks) with chunks[j] = the bookkeeping of output irrep j."""
    irreps_sh, irreps_out = Irreps(irreps_sh), Irreps(irreps_out)
    gl = PlanarLayout(irreps_out)
    chunks = []
    for b in branches:
        for sp in _tp_superpaths(b["nsrc"], b["lay"], irreps_sh, irreps_out, None if b["tp_w"] is None else np.asarray(b["tp_w"]),
                                 np.asarray(b["w3"]) / math.sqrt(H), np.asarray(b["ls_w"]), None if b["lo_w"] is None else np.asarray(b["lo_w"]),
                                 bool(b.get("uvu", False))):
            nc = 2 * sp["mm"] + 1
            step = min(rtm_max(nc) * 16, seg_rows_cap(sp["lk"]))
            for r0 in range(0, sp["nmid"], step):
                chunks.append(dict(sp=sp, branch=b["name"], nsrc=b["nsrc"], srcs=b["srcs"], lay=b["lay"], mlp=b["mlp"], r0=r0,
                                   r1=min(sp["nmid"], r0 + step)))
    out_irreps = Irreps([(c["r1"] - c["r0"], c["sp"]["lk"], c["sp"]["pk"]) for c in chunks])
    progs = []
    for which in ("A", "B"):
        prog, seg_of = new_program(out_irreps, H)
        for j, c in enumerate(chunks):
            sp, r0, r1 = c["sp"], c["r0"], c["r1"]
            n, mm, lk, mk = r1 - r0, sp["mm"], sp["lk"], sp["mk"]
            nc = 2 * mm + 1
            rtm = ceil_div(n, 16)
            rho = np.arange(n)
            phys = 16 * (rho // 16) + 4 * (rho % 4) + (rho % 16) // 4        # see add_tp_items
            R = rtm * 16
            seg = seg_of[j]
            rto = prog.segs[seg][2]
            w3p = np.zeros((H, R))
            w3p[0, phys] = 1.0                                 # radial scale 1: the launch gets hidden rows with a 1 in column 0
            w3_off = prog.add_weights(_frag_A(w3p, prog.hidden_pad // 4, rtm, True))
            cfp = np.zeros((R, nc))
            cfp[phys] = sp["cf"][r0:r1] if which == "A" else 1.0
            cf_off = prog.add_weights(_cf_block(cfp, rtm, nc))
            Ip = np.zeros((R, rto * 16))
            Ip[phys, rho] = 1.0                                # GEMM2 = identity: output channel = logical row
            a2_off = prog.add_weights(Ip.reshape(rtm, 4, 4, rto, 16).transpose(3, 0, 1, 4, 2).reshape(rto, rtm, 64, 4))
            if which == "A":
                lay, mi, i = c["lay"], sp["mi"], sp["i"]
                ksteps = lay.mulp[i] // 4
                x4 = use_x4(lay.mulp[i], nc)
                a1 = []
                for s_ in range(c["nsrc"]):
                    Wk = np.zeros((mi, R))
                    Wk[:, phys] = sp["W"][r0:r1, s_ * mi:(s_ + 1) * mi].T
                    a1.append(_frag_A(Wk, ksteps, rtm, x4))
                a1_off = prog.add_weights(np.stack(a1))
                _add_item(prog, seg, IT_TP, list(c["srcs"]), lay.off[i], lay.mulp[i], sp["li"], mm, sp["par"], ksteps, rtm, c["mlp"],
                          a1_off, w3_off, cf_off, a2_off, n, nk2=ceil_div(n, 4))
            else:
                k = sp["k"]
                ksteps = gl.mulp[k] // 4
                Lk = np.zeros((mk, R))
                Lk[:, phys] = sp["L"][r0:r1].T
                a1_off = prog.add_weights(_frag_A(Lk, ksteps, rtm, use_x4(gl.mulp[k], nc))[None])
                _add_item(prog, seg, IT_TP, [0], gl.off[k], gl.mulp[k], lk, mm, 0, ksteps, rtm, c["mlp"],
                          a1_off, w3_off, cf_off, a2_off, n, nk2=ceil_div(n, 4))
        progs.append(prog.finalize())
    lay_out = PlanarLayout(out_irreps)
    for j, c in enumerate(chunks):
        c["out_off"], c["out_mulp"] = lay_out.off[j], lay_out.mulp[j]
    return progs[0], progs[1], chunks


# ------------------------------------------------------------------------------------------------ fused weight-gradient kernel (csrc/tp_wgrad.hip)
WG_UNIT_I32 = 64            # ints per unit record, see WgFused
WG_WREC, WG_WREC_I32 = 24, 10
WG_WAVES = 4
WG_LDS_ROW_MAX = 640        # floats: 2 buffers x 16 rows x 640 x 4 B = 80 KB (two workgroups per CU)
WG_MAX_PIECES = 10          # float4 pieces a thread holds in flight while the next edge tile is staged (csrc/tp_wgrad.hip WG_NP) ...


def wg_pieces_of_nc(nc: int) -> int:
    """... by the column count of a wave's row tile (csrc/tp_wgrad.hip WG_NP_OF); a unit's operand tile must fit its most demanding wave"""
    return WG_MAX_PIECES if nc <= 9 else 5


def wg_shape_ok(nc: int, g1: int, g2: int) -> bool:
    """template instantiations of csrc/tp_wgrad.hip (WG_CASES_*): columns x tiles of 16 input / output channels"""
    if nc == 1:
        return 1 <= g1 <= 4 and 1 <= g2 <= 4
    if nc in (3, 5, 7):
        return 1 <= g1 <= 2 and 1 <= g2 <= 2
    return nc in (9, 11, 13) and g1 == 1 and g2 == 1


@dataclass
class WgFused:
    """Launch tables of the fused weight-gradient kernel for the weighted tensor-product branches of one block.
    A UNIT = up to four 16-row tiles of super-paths (i, k) that read the SAME input irrep i of one branch (any k); a workgroup owns one unit
    and a range of edge tiles; wave w works on its row tile for the edge tile `et` of every iteration (ET edge tiles per iteration: units with
    fewer than four row tiles put several edge tiles side by side), weights and accumulators stay in its registers:
        g_W[u, row] += sum_{c, e} x[e, u, comp(c)] * (s cf (L g))[e, row, c]        (K = the 16 edges: the C fragment of the first-stage MFMAs IS the B operand)
        g_L[w, row] += sum_{c, e} g[e, w, col(c)] * (s cf (W x))[e, row, c]
        gs[e, ch(row)] = sum_c cf (W x) (L g)                                         (written per edge: last radial layer / hidden-layer gradients)
    units [n, WG_UNIT_I32] int32:
         0 nsrc   1 slot0   2 slot1   3 x_off (floats into a source row: first staged component of irrep i)   4 in_mulp   5 x pieces per source
         6 number of gradient spans   7 mlp (which hidden rows / which gs buffer)   8 ET   9 RS (LDS row stride, floats, == 4 mod 64)   10 RS / 4
        11 h pieces   12 G1 = ceil(in_mulp / 16)   13 cost (MFMAs of the dearest wave per iteration)   14 busy waves   15 branch
        16 + 2 s, 17 + 2 s: gradient span s (s < 4): float offset into a gradient row, pieces
        24 + 10 w ...: wave w: busy, et, nc, par (1: column c reads component nc-1-c of its span), x column offset (floats, from the staged span),
                       LDS offset of its gradient span (floats), g_mulp, weight offset (floats), accumulator offset (floats, per split), chtab offset
    weights per row tile: [W: nsrc x G1 x 64 x 4][L: G2 x 64 x 4][W3: H/16 x 64 x 4][cf: nc x 16], B-operand fragments in natural-K order
    (lane (row, kk) holds M[row][4 (4 G + q) + kk], q = float4 component).  accumulators per (split, wave): [nsrc*G1 + G2 fragments][16 rows][16 channels]."""
    units: np.ndarray
    weights: np.ndarray
    chtab: np.ndarray
    acc_floats: int
    hidden: int
    branch_names: List[str]
    nch: List[int]
    tp_pos: List[Optional[np.ndarray]]      # per branch: [tp_size, 4] positions in the accumulator block (acc_floats - 1 = a zero slot)
    tp_scale: List[Optional[np.ndarray]]
    l_pos: List[np.ndarray]                 # per branch: [ls_size, 4]
    lds_bytes: int
    gs_complete: bool = False               # every radial channel of every branch is written by some row tile (gs needs no zero fill)
    ch_ranges: Optional[List[List[Tuple[int, int]]]] = None      # per branch: the contiguous ranges of radial channels some row tile writes (the others stay 0)
    mfma_per_tile: float = 0.0              # issued MFMAs per 16 edges, all units
    bytes_per_edge: float = 0.0             # staged bytes per edge, all units


@single_thread_blas
def build_tp_wgrad_fused(branches, irreps_sh, irreps_out, H: int, zero_inputs: Optional[Dict[str, Sequence[int]]] = None) -> WgFused:
    """see WgFused; branches as build_tp_wgrad_programs (weighted branches only).  zero_inputs: {branch name: input irreps whose rows are structurally
    zero} -- every gradient a super-path that reads one of them feeds (g_W = x^T ..., g_L = g^T (s cf W x), gs = sum cf (W x)(L g)) is exactly zero, so its
    row tiles are not built: the parameters keep the zero slot, the radial channels stay at the zero fill (gs_complete is False then)."""
    zero_inputs = {k: set(int(i) for i in v) for k, v in (zero_inputs or {}).items()}
    irreps_sh, irreps_out = Irreps(irreps_sh), Irreps(irreps_out)
    gl = PlanarLayout(irreps_out)
    if H != 64:
        raise NotImplementedError("fused weight gradients: hidden width of the radial MLP must be 64")
    units, wparts, chparts = [], [], []
    woff = accoff = choff_t = 0
    tp_pos, tp_scale, l_pos, nchs = [], [], [], []
    seen_ch: List[set] = []
    lds_max = 0
    total_cost = total_bytes = 0.0
    for bi, b in enumerate(branches):
        if b["tp_w"] is None:
            raise NotImplementedError("fused weight gradients: unweighted (uvu) branches carry no tensor-product weights")
        lay, nsrc = b["lay"], b["nsrc"]
        w3 = np.asarray(b["w3"], dtype=np.float64) / math.sqrt(H)
        tp_size, ls_size = int(np.asarray(b["tp_w"]).size), int(np.asarray(b["ls_w"]).size)
        tpp_t, tpp_p, lpp_t, lpp_p = [], [], [], []            # (flat parameter index, accumulator position) pairs, one per wave that feeds it
        tps = np.zeros(tp_size)
        nchs.append(int(w3.shape[1]))
        by_i: Dict[int, list] = {}
        for sp in _tp_superpaths(nsrc, lay, irreps_sh, irreps_out, np.asarray(b["tp_w"]), w3, np.asarray(b["ls_w"]),
                                 None if b["lo_w"] is None else np.asarray(b["lo_w"]), False):
            if sp["i"] in zero_inputs.get(b["name"], ()):
                continue
            nc = 2 * sp["mm"] + 1
            g1, g2 = ceil_div(lay.mulp[sp["i"]], 16), ceil_div(gl.mulp[sp["k"]], 16)
            if not wg_shape_ok(nc, g1, g2):
                raise NotImplementedError(f"fused weight gradients: no kernel instantiation for {nc} columns x {g1} / {g2} channel tiles")
            cost = nc * (nsrc * (lay.mulp[sp["i"]] // 4) + gl.mulp[sp["k"]] // 4) + H // 4 + 4 * nc * (nsrc * g1 + g2)      # MFMAs per 16 edges and row tile
            for t in range(ceil_div(sp["nmid"], 16)):
                by_i.setdefault(sp["i"], []).append((sp, t, cost))
        for i, tiles in by_i.items():
            tiles.sort(key=lambda x: -x[2])                    # like-priced row tiles share a workgroup (its waves meet at a barrier every iteration)
            in_mulp, li = lay.mulp[i], lay.irreps[i][1]
            q0 = 0
            while q0 < len(tiles):
                # greedily take up to four row tiles whose operand row fits the LDS budget
                take, segs = [], []
                while q0 + len(take) < len(tiles) and len(take) < WG_WAVES:
                    sp = tiles[q0 + len(take)][0]
                    segs2 = segs if sp["k"] in segs else segs + [sp["k"]]
                    mmax = max([sp["mm"]] + [t_[0]["mm"] for t_ in take])
                    used = nsrc * (2 * mmax + 1) * in_mulp + sum((2 * min(li, irreps_out[k][1]) + 1) * gl.mulp[k] for k in segs2) + H
                    RS = used + ((4 - used) % 64)
                    if RS > WG_LDS_ROW_MAX or ceil_div(used // 4, 16) > wg_pieces_of_nc(2 * mmax + 1):      # 16 threads stage one row
                        break
                    take.append(tiles[q0 + len(take)])
                    segs = segs2
                if not take:
                    raise NotImplementedError("fused weight gradients: a 16-edge operand tile does not fit the LDS budget")
                q0 += len(take)
                mmax = max(t_[0]["mm"] for t_ in take)
                xp = (2 * mmax + 1) * in_mulp // 4
                seg_p = [(2 * min(li, irreps_out[k][1]) + 1) * gl.mulp[k] // 4 for k in segs]
                used = 4 * (nsrc * xp + sum(seg_p)) + H
                RS = used + ((4 - used) % 64)                   # == 4 mod 64: conflict-free dword reads of 16 rows x 4 K-slots AND of 4 rows x 16 channels
                nt = len(take)
                ET = max(et_ for et_ in (1, 2, 4) if et_ == 1 or (et_ * nt <= WG_WAVES and et_ * RS <= WG_LDS_ROW_MAX
                                                                  and ceil_div(used // 4 * et_, 16) <= wg_pieces_of_nc(2 * mmax + 1)))      # 16 / ET threads per row
                rec = [0] * WG_UNIT_I32
                rec[0:16] = [nsrc, b["srcs"][0], b["srcs"][-1], lay.off[i] + (li - mmax) * in_mulp, in_mulp, xp, len(segs), b["mlp"], ET, RS, RS // 4, H // 4,
                             ceil_div(in_mulp, 16), int(max(t_[2] for t_ in take)), nt * ET, bi]
                seg_lds, o = {}, 4 * nsrc * xp
                for s_, k in enumerate(segs):
                    lk = irreps_out[k][1]
                    rec[16 + 2 * s_] = gl.off[k] + (lk - min(li, lk)) * gl.mulp[k]
                    rec[17 + 2 * s_] = seg_p[s_]
                    seg_lds[k] = o
                    o += 4 * seg_p[s_]
                for w_ in range(nt * ET):
                    sp, t, cost = take[w_ % nt][:3]
                    e = w_ // nt
                    k, mi, mk, mm, lk = sp["k"], sp["mi"], sp["mk"], sp["mm"], sp["lk"]
                    nc, g_mulp = 2 * mm + 1, gl.mulp[k]
                    G1, G2 = ceil_div(in_mulp, 16), ceil_div(g_mulp, 16)
                    r0, r1 = 16 * t, min(sp["nmid"], 16 * t + 16)
                    n = r1 - r0
                    nfr = nsrc * G1 + G2
                    if e == 0:                                 # the row tile's weights and channel table (shared by its edge-tile copies)
                        Wp = np.zeros((nsrc, in_mulp, 16))
                        for s_ in range(nsrc):
                            Wp[s_, :mi, :n] = sp["W"][r0:r1, s_ * mi:(s_ + 1) * mi].T
                        Lp = np.zeros((g_mulp, 16))
                        Lp[:mk, :n] = sp["L"][r0:r1].T
                        W3p = np.zeros((H, 16))
                        W3p[:, :n] = w3[:, sp["ch"][r0:r1]]
                        cfp = np.zeros((nc, 16))
                        cfp[:, :n] = sp["cf"][r0:r1].T
                        blob = np.concatenate([np.stack([_frag_A(Wp[s_], in_mulp // 4, 1, False) for s_ in range(nsrc)]).reshape(-1),
                                               _frag_A(Lp, g_mulp // 4, 1, False).reshape(-1), _frag_A(W3p, H // 4, 1, False).reshape(-1), cfp.reshape(-1)])
                        ch = np.full(16, -1, np.int64)
                        ch[:n] = sp["ch"][r0:r1]
                        wparts.append(blob)
                        chparts.append(ch)
                        take[w_ % nt] = (sp, t, cost, woff, choff_t)
                        woff += blob.size
                        choff_t += 16
                    my_w, my_ch = take[w_ % nt][3:]
                    rec[WG_WREC + WG_WREC_I32 * w_:WG_WREC + WG_WREC_I32 * (w_ + 1)] = [1, e, nc, sp["par"], (mmax - mm) * in_mulp, seg_lds[k], g_mulp, my_w, accoff, my_ch]
                    # where the gradient of every flat parameter lands: this wave's block, fragment f, row, channel
                    meta = sp["meta"][r0:r1]
                    rho = np.arange(n)
                    base_w = np.array([sp["woff"][m_[0]] + m_[1] for m_ in meta], dtype=np.int64)
                    cps = np.array([m_[2] for m_ in meta])
                    lrows = np.array([m_[3] for m_ in meta], dtype=np.int64)
                    u = np.arange(mi)
                    for s_ in range(nsrc):
                        pos = accoff + ((s_ * G1 + u // 16) * 256 + u % 16)[None, :] + rho[:, None] * 16
                        tgt = base_w[:, None] + ((s_ * mi + u) * mk)[None, :]
                        tpp_t.append(tgt.reshape(-1))
                        tpp_p.append(pos.reshape(-1))
                        tps[tgt] = cps[:, None]
                    w2 = np.arange(mk)
                    pos = accoff + ((nsrc * G1 + w2 // 16) * 256 + w2 % 16)[None, :] + rho[:, None] * 16
                    tgt = sp["lin"][0] + lrows[:, None] * mk + w2[None, :]
                    lpp_t.append(tgt.reshape(-1))
                    lpp_p.append(pos.reshape(-1))
                    accoff += nfr * 256
                    if e == 0:
                        total_cost += cost
                units.append(rec)
                lds_max = max(lds_max, 2 * ET * 16 * RS * 4)
                total_bytes += used * 4.0
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)      # (a branch whose every input irrep is structurally zero has no row tile)
        tp_pos.append((cat(tpp_t), cat(tpp_p), tp_size))
        tp_scale.append(tps)
        l_pos.append((cat(lpp_t), cat(lpp_p), ls_size))
        seen_ch.append(set(int(c) for sp_ in by_i.values() for (sp__, t_, *_) in sp_ for c in sp__["ch"][16 * t_:16 * t_ + 16]))
    zero = accoff                                              # one spare slot that stays zero
    def table(pairs):                                          # [n parameters, 4]: the (<= 4) edge-tile copies of every parameter's slot, padded with the zero slot
        tgt, pos, n_ = pairs
        order_ = np.argsort(tgt, kind="stable")
        tgt, pos = tgt[order_], pos[order_]
        first = np.searchsorted(tgt, tgt, side="left")
        col = np.arange(tgt.size) - first
        assert col.max(initial=0) < 4
        out = np.full((n_, 4), zero, np.int64)
        out[tgt, col] = pos
        return out
    def ranges(chs):
        out = []
        for c in sorted(chs):
            if out and out[-1][1] == c:
                out[-1][1] = c + 1
            else:
                out.append([c, c + 1])
        return [tuple(r) for r in out]
    U = np.asarray(units, dtype=np.int64)
    order = np.argsort(-U[:, 13], kind="stable")               # dearest units first (the hardware hands workgroups out in order)
    return WgFused(units=U[order].astype(np.int32), weights=np.concatenate(wparts), chtab=np.concatenate(chparts).astype(np.int32), acc_floats=accoff + 1,
                   hidden=H, branch_names=[b["name"] for b in branches], nch=nchs, tp_pos=[table(t) for t in tp_pos], tp_scale=tp_scale,
                   l_pos=[table(t) for t in l_pos], lds_bytes=lds_max, gs_complete=all(s_ == set(range(n_)) for s_, n_ in zip(seen_ch, nchs)),
                   ch_ranges=[ranges(s_) for s_ in seen_ch], mfma_per_tile=total_cost, bytes_per_edge=total_bytes)


def message_pack_wgrad_branches(sd: Dict[str, np.ndarray], irreps_node, irreps_edge):
    """the two weighted branches of a non-lite MessagePackBlock, with the reference's parameter names (message_passing.py:112-160)"""
    out = []
    for name, nsrc, srcs, irr, mlp in (("node", 2, [SRC_XS, SRC_XD], irreps_node, 0), ("edge", 1, [SRC_F], irreps_edge, 1)):
        keys = dict(tp=f"{name}_tensor_product.weight", ls=f"{name}_linear_scaler.linear_out.weight", lo=f"{name}_linear_out.weight",
                    gen=f"{name}_weight_generator")
        _, w3 = _last_layer(sd, keys["gen"])
        out.append(dict(name=name, nsrc=nsrc, srcs=srcs, lay=PlanarLayout(Irreps(irr)), mlp=mlp, keys=keys, tp_w=sd[keys["tp"]], w3=w3,
                        ls_w=sd[keys["ls"]], lo_w=sd[keys["lo"]]))
    return out


def embedding_wgrad_branches(sd: Dict[str, np.ndarray], num_types: int, lite_mode: bool = False):
    """the single branch of PairInteractionEmbeddingBlock.conv_tp (embeddings.py:328-334): input num_types x 0e; lite_mode: the
    unweighted uvu product (no tensor_product.weight, one radial weight per INPUT channel and path)"""
    keys = dict(tp=None if lite_mode else "tensor_product.weight", ls="linear_scaler.linear_out.weight", lo=None, gen="weight_generator")
    _, w3 = _last_layer(sd, keys["gen"])
    return [dict(name="emb", nsrc=1, srcs=[SRC_XS], lay=PlanarLayout([(num_types, 0, 1)]), mlp=0, keys=keys, tp_w=None if lite_mode else sd[keys["tp"]],
                 w3=w3, ls_w=sd[keys["ls"]], lo_w=None, uvu=lite_mode)]


def embedding_wgrad_branches_split(sd: Dict[str, np.ndarray], num_types: int):
    """the embedding TP's branch for the FUSED weight-gradient kernel (late r5): its num_types x 0e input is wider than the four 16-channel tiles a wave of
    csrc/tp_wgrad.hip holds, so the row is presented as TWO sources of num_types / 2 channels -- exactly the layout of a MessagePackBlock's node branch
    ((2 mul) x ir = sender channels, then receiver channels): same flat parameter indices, same fan-in.  num_types / 2 must be a multiple of 4."""
    if num_types % 8:
        raise NotImplementedError("fused weight gradients of the embedding TP: num_types must be a multiple of 8")
    b = embedding_wgrad_branches(sd, num_types, False)[0]
    return [dict(b, nsrc=2, srcs=[0, 1], lay=PlanarLayout([(num_types // 2, 0, 1)]))]


@single_thread_blas
def build_message_pack_wgrad_programs(sd: Dict[str, np.ndarray], irreps_node, irreps_edge, irreps_sh, irreps_out):
    br = message_pack_wgrad_branches(sd, irreps_node, irreps_edge)
    return build_tp_wgrad_programs(br, irreps_sh, irreps_out, br[0]["w3"].shape[0])

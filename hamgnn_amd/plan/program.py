"""The Program container of the fused edge kernels and the packing of its items: MFMA fragment order of the A operands, packed CG coefficients, segments, the
tensor-product / adjoint / plain-Linear item builders (hamgnn_amd.plan: see the package docstring for the math and the lane conventions)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import so3
from ..so3 import Irreps
from .layout import ITEM_I32, IT_LIN, IT_POST, IT_TP, PlanarLayout, SEG_I32, STAGE_FLOATS, ceil_div, rtm_max, tp_instructions

# ------------------------------------------------------------------------------------------------ program container


# dtype of the packed weight blob: float32 for the device; hamgnn_amd/repack.py probes the builders in float64 (probe_dtype)
_WEIGHT_DTYPE = [np.float32]


class probe_dtype:
    """with probe_dtype(): the builders keep their weight blobs in float64 (used to discover blob = const + coef * source[idx])"""

    def __enter__(self):
        _WEIGHT_DTYPE[0] = np.float64

    def __exit__(self, *a):
        _WEIGHT_DTYPE[0] = np.float32


@dataclass
class Program:
    out_layout: PlanarLayout
    hidden: int = 0                                   # radial hidden width H or 0

    @property
    def hidden_pad(self):                             # H padded to the permuted-K granule (16)
        return ceil_div(self.hidden, 16) * 16

    segs: List[List[int]] = field(default_factory=list)
    seg_items: List[List[List[int]]] = field(default_factory=list)    # per segment: item records (kept contiguous per segment)
    chunks: List[np.ndarray] = field(default_factory=list)
    _woff: int = 0
    tile_floats: int = 0                              # dynamic LDS floats per workgroup (4 wave-private tiles)
    flops_per_row: float = 0.0                        # algorithmic (unpadded) flops per edge/row
    mfma_per_wave: int = 0                            # issued MFMAs per 16-row wave tile (padded)
    mfma_radial: int = 0                              # ... of which the radial scales S = W3^T h, counted as fp32 MFMAs of K = 4 (the kernel issues 6 half-precision ones per row tile instead of H / 4)
    mfma_odd_skipped: int = 0                         # of those, the centre-column MFMAs of odd items that csrc/tp_is.hip does not issue
    # merged items (input-stationary kernel only): an item whose GEMM2 rows span SEVERAL output segments.  vsegs[v] = the member
    # segments in row order; such an item is filed under its first member, carries v + 1 in its row_off field (item[16]) and its L'
    # fragments address the concatenated channels of the members.  seg_key[s] = the segment whose work group owns segment s's tile.
    vsegs: List[List[int]] = field(default_factory=list)
    seg_key: Dict[int, int] = field(default_factory=dict)
    # r6: radial-scale operands in SPLIT HALF PRECISION (see w3_split_fill): (float offset of a W3 fragment block, row tiles); its split twin follows it directly
    w3_regions: List[Tuple[int, int]] = field(default_factory=list)

    def add_weights(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr, dtype=_WEIGHT_DTYPE[0]).reshape(-1)
        off = self._woff
        self.chunks.append(arr)
        self._woff += arr.size
        pad = (-self._woff) % 4                       # keep 16-byte alignment of every operand block
        if pad:
            self.chunks.append(np.zeros(pad, _WEIGHT_DTYPE[0]))
            self._woff += pad
        return off

    def finalize(self):
        self.weights = np.concatenate(self.chunks) if self.chunks else np.zeros(4, np.float32)
        if self.weights.dtype == np.float32:                   # (probe_dtype runs leave the split twins zero: they are not affine in the sources, repack.py fills them on the device)
            self.w3_exp = w3_split_exp(self.weights, self.w3_regions)
            self.w3_split_ok = w3_split_fill(self.weights, self.w3_regions, self.w3_exp)
        items = []
        for seg, lst in zip(self.segs, self.seg_items):
            seg[5] = len(items)
            items += lst
            seg[6] = len(items)
        self.seg_table = np.asarray(self.segs, dtype=np.int32).reshape(-1, SEG_I32)
        self.item_table = np.asarray(items, dtype=np.int32).reshape(-1, ITEM_I32)
        del self.chunks
        return self


# ---- the radial scale S = W3^T h on v_mfma_f32_16x16x32_f16 with SPLIT operands (r6, csrc/tp_is.hip) ----------------------------------------
# 27.5 % of a MessagePackBlock's MFMAs compute the per-edge radial scales S[(l_sh, w), e] = sum_h W3[h, row] h2[e, h] (K = the 64 hidden units).  On the
# fp32 matrix pipe that is 16 MFMAs of 32 cycles per 16-row tile.  With every operand written as a pair of halves,
#       x 2^s = hi + 2^-11 lo,      hi = f16(x 2^s),   lo = f16((x 2^s - hi) 2^11)                       (22 significant bits)
# S 2^(sw + sh) = W_hi h_hi + 2^-11 (W_hi h_lo + W_lo h_hi) is 3 half-precision MFMAs of K = 32 per half of the hidden units: 6 MFMAs of 16 cycles per row
# tile, fp32 accumulation in two chains, same operand bytes; the dropped lo * lo term is 2^-22 of a product.  Measured end to end on the emulator (Si2, set-A,
# 3 layers): + 6e-7 relative on H (profiles/r06_tp_is.md).  Why the scalings: the half-precision MFMAs FLUSH SUBNORMAL INPUTS -- an unscaled lo = f16(x - hi)
# of a weight around 0.1 is below 2^-14 and vanishes (measured: 9e-5 on H); with the remainder scaled by 2^11 the lo terms live in the range of the hi terms,
# and 2^sw (per program: the largest |W3| lands in [2^12, 2^13)) / 2^sh (the kernel's constant for the hidden rows) keep small values normal.
# The weights are split HERE (no VALU work in the kernel): every W3 fragment block [G = 4][rt][lane][q] (fp32, read by the segment-stationary kernel, the
# emulators and the hidden != 64 path) is followed by its twin [t = 2][rt][term = hi, lo][lane][4 dwords], dword d of lane (g, i) = the K-slots (2d, 2d + 1)
# of half t packed as two f16 (low half-word first), slot s = 4 (G - 2t) + q: the SAME (lane, slot) -> hidden unit map as the fp32 fragments, so the kernel's
# resident hidden rows pair up with it without a shuffle.  Only for 64 (padded) hidden units.
W3_SPLIT_MAX = 6.0e4          # a scaled weight above this is beyond the half-precision range: the launches keep the fp32 form (Program.w3_split_ok, ops.check_w3_split)
SPLIT_LO_EXP = 11             # the remainder is scaled by 2^11 before it is rounded to half precision
SPLIT_H_EXP = 6               # sh: the kernel scales the hidden rows by 2^6 (|h| < 1023 stays finite; hidden activations are O(1))


def w3_split_index(rtm: int):
    """(source float offsets [n, 2], term-major destination dword offsets [2][n]) of ONE block, relative to the block's / its twin's start"""
    t, rt, lane, d = np.meshgrid(np.arange(2), np.arange(rtm), np.arange(64), np.arange(4), indexing="ij")
    src = lambda s_: (((2 * t + s_ // 4) * rtm + rt) * 64 + lane) * 4 + s_ % 4
    even, odd = src(2 * d), src(2 * d + 1)
    dst = lambda term: (((t * rtm + rt) * 2 + term) * 64 + lane) * 4 + d
    return np.stack([even.reshape(-1), odd.reshape(-1)], 1), np.stack([dst(0).reshape(-1), dst(1).reshape(-1)])


def f16_split(x: np.ndarray, exp: int = 0):
    """x (float32) -> (hi, lo) float16 with x 2^exp = hi + 2^-11 lo to 22 bits: hi = f16(x 2^exp), lo = f16((x 2^exp - hi) 2^11)"""
    x = np.asarray(x, dtype=np.float32) * np.float32(2.0 ** exp)
    with np.errstate(over="ignore", invalid="ignore"):         # (beyond the half-precision range: inf -- such a program keeps the fp32 form, w3_split_fill)
        hi = x.astype(np.float16)
        lo = ((x - hi.astype(np.float32)) * np.float32(2.0 ** SPLIT_LO_EXP)).astype(np.float16)
    return hi, lo


def w3_split_exp(weights: np.ndarray, regions) -> int:
    """sw of a program: the largest |W3| of its blocks lands in [2^12, 2^13) (a factor 8 below the half-precision maximum: room for an optimiser's steps)"""
    mx = max((float(np.abs(weights[off:off + 4 * rtm * 256]).max(initial=0.0)) for off, rtm in regions), default=0.0)
    if not np.isfinite(mx) or mx <= 0.0:
        return 0
    return int(np.clip(12 - int(np.floor(np.log2(mx))), -40, 40))


def w3_split_fill(weights: np.ndarray, regions, exp: int) -> bool:
    """write the split twins of all W3 fragment blocks of a float32 blob, in place; False: a scaled weight is beyond the half-precision range (the launches then
    keep the fp32 form: part record [12] = 0)"""
    ok = True
    for off, rtm in regions:
        n = 4 * rtm * 256
        src, dst = w3_split_index(rtm)
        blk = weights[off:off + n]
        hi_e, lo_e = f16_split(blk[src[:, 0]], exp)
        hi_o, lo_o = f16_split(blk[src[:, 1]], exp)
        twin = weights[off + n:off + 2 * n].view(np.uint32)
        twin[dst[0]] = hi_e.view(np.uint16).astype(np.uint32) | (hi_o.view(np.uint16).astype(np.uint32) << 16)
        twin[dst[1]] = lo_e.view(np.uint16).astype(np.uint32) | (lo_o.view(np.uint16).astype(np.uint32) << 16)
        ok = ok and bool(np.abs(blk).max(initial=0.0) * 2.0 ** exp <= W3_SPLIT_MAX)
    return ok


def _add_w3(prog: "Program", w3p: np.ndarray, rtm: int) -> int:
    """the W3 fragment block of an item (+ its split twin when the program has 64 hidden units); returns the block's offset"""
    off = prog.add_weights(_frag_A(w3p, prog.hidden_pad // 4, rtm, True))
    if prog.hidden_pad == 64:
        prog.add_weights(np.zeros(4 * rtm * 256))
        prog.w3_regions.append((off, rtm))
    return off


def _frag_A(mat_kxr: np.ndarray, ksteps: int, rtm: int, x4: bool) -> np.ndarray:
    """mat[k, row] -> A fragments [ngrp][rtm][64 lanes][4]: one float4 per lane covers 4 MFMA K-steps (q = 0..3).
    lane L = (i = L&15, g = L>>4) holds mat[k(G, q, g)][16 rt + i] with
        x4 (permuted K, B operand loaded as float4):  k = 16 G + 4 g + q
        x1 (B operand loaded as dwords)            :  k = 4 (4 G + q) + g
    zero padded to ngrp = ceil(ksteps / 4) groups."""
    K, Rr = mat_kxr.shape
    ngrp = ceil_div(ksteps, 4)
    P = np.zeros((ngrp * 16, rtm * 16), dtype=np.float64)
    P[:K, :Rr] = mat_kxr
    P = P.reshape(ngrp, 4, 4, rtm, 16)                      # x4: [G, g, q, rt, i] ; x1: [G, q, g, rt, i]
    if x4:
        return P.transpose(0, 3, 1, 4, 2).reshape(ngrp, rtm, 64, 4)
    return P.transpose(0, 3, 2, 4, 1).reshape(ngrp, rtm, 64, 4)


def _cf_block(cfp: np.ndarray, rtm: int, nc: int) -> np.ndarray:
    """CF operand of a tensor-product item, two forms back to back: [rt][c][g][r] (segment-stationary kernel, emulators) and the PACKED
    form the input-stationary kernel reads (csrc/tp_is.hip): the pairs p = rt * nc + c in groups of 16 as [J][g][p % 16][r] -- lane
    (g, p) of a wave holds the float4 of pair p after ONE load per 16 pairs (instead of one load and four registers per pair); the scale
    step broadcasts it along the 16 lanes of row g by DPP (row_newbcast).  The packed block starts rtm * nc * 16 floats behind item[13]."""
    old = cfp.reshape(rtm, 4, 4, nc).transpose(0, 3, 1, 2)             # [rt][c][g][r]
    npair = rtm * nc
    pk = np.zeros((ceil_div(npair, 16) * 16, 4, 4), dtype=old.dtype)
    pk[:npair] = old.reshape(npair, 4, 4)
    pk = pk.reshape(-1, 16, 4, 4).transpose(0, 2, 1, 3)                # [J][g][p][r]
    return np.concatenate([old.reshape(-1), pk.reshape(-1)])


def use_x4(in_mulp, nc):
    """permuted-K float4 B loads: channel block a multiple of 16 and few columns (register budget of the kernel)."""
    return in_mulp % 16 == 0 and nc <= 3


def _add_segment(prog: Program, lk, mul_k, out_index, flags):
    lay = prog.out_layout
    if lk > 6:
        raise NotImplementedError(f"output irreps with l = {lk} > 6 have no kernel epilogue instantiation")
    assert mul_k <= seg_rows_cap(lk)
    rto = ceil_div(mul_k, 16)
    prog.segs.append([lk, mul_k, rto, lay.off[out_index], lay.mulp[out_index], 0, 0, flags])
    prog.seg_items.append([])
    prog.tile_floats = max(prog.tile_floats, 4 * ((mul_k + 1) * ((2 * lk + 1) * 16 + 4) + STAGE_FLOATS))   # 4 waves x (tile [mul_k rows + trash row] + DMA ring)
    return len(prog.segs) - 1


# (MM, RTM) template instantiations of both fused kernels (csrc/tp_fused.hip HG_CASE / csrc/tp_is.hip IS_CASE): an item outside this
# set would be skipped silently by the kernels' dispatch, so the planner refuses to emit one
KERNEL_RTM_MAX = (4, 4, 3, 2, 2, 1, 1)


def _add_item(prog: Program, seg, typ, srcs, in_off, in_mulp, li, mm, neg, ksteps, rtm, mlp, a1, w3, cf, a2, nrows, row_off=0, nk2=None, rto=None):
    assert len(srcs) in (1, 2)
    if typ != IT_POST and not (0 <= mm < len(KERNEL_RTM_MAX) and 1 <= rtm <= KERNEL_RTM_MAX[mm]):
        raise NotImplementedError(f"no kernel instantiation for an item with min(l_in, l_out) = {mm} and {rtm} row tiles")
    nk2 = 4 * rtm if nk2 is None else nk2                      # GEMM2 K-steps actually issued (item[18])
    assert 4 * (rtm - 1) < nk2 <= 4 * rtm                      # only the last row tile holds K-steps that are not issued (csrc/tp_is.hip:IS_NK2_OK)
    if (2 * mm + 1) * in_mulp > 160:
        raise NotImplementedError(f"input irrep block too wide for the kernel's B staging ring: (2*{mm}+1) x {in_mulp} channels > 160")
    rec = [typ, srcs[0], srcs[1] if len(srcs) == 2 else -1, in_off, in_mulp, li, mm, neg, ksteps, rtm, mlp,
           a1, w3, cf, a2, nrows, row_off, 1 if use_x4(in_mulp, 2 * mm + 1) else 0, nk2, seg]
    assert len(rec) == ITEM_I32
    prog.seg_items[seg].append(rec)
    nc = 2 * mm + 1
    rto = prog.segs[seg][2] if rto is None else rto
    n = len(srcs) * ksteps * rtm * nc
    if typ == IT_POST:
        n = (prog.hidden_pad // 4) * rto + rto * rto * 4 * (2 * prog.segs[seg][0] + 1)
    if typ == IT_TP:
        n += (prog.hidden_pad // 4) * rtm + rto * nk2 * nc
        prog.mfma_radial += (prog.hidden_pad // 4) * rtm
        if neg and mm > 0:                                     # odd super-path: the input-stationary kernel skips the (zero) centre column
            prog.mfma_odd_skipped += len(srcs) * ksteps * rtm + rto * nk2
    prog.mfma_per_wave += n


# ------------------------------------------------------------------------------------------------ builders


def linear_scaler_layout(nsrc: int, in_layout: PlanarLayout, irreps_sh, irreps_out):
    """[(k, offset into LinearScaleWithWeights.linear_out.weight, fan_in, offset into the trailing o3.Linear(out -> out) weight, mul_k)]
    of a weighted ("uvw") tensor-product branch, in the order the flat linear_scaler weight stores its blocks"""
    irreps_sh, irreps_out = Irreps(irreps_sh), Irreps(irreps_out)
    irr_in = Irreps([(m * nsrc, l, p) for m, l, p in in_layout.irreps])
    by_k: Dict[int, List[int]] = {}
    for n, (i, j, k, slot) in enumerate(tp_instructions(irr_in, irreps_sh, irreps_out)):
        by_k.setdefault(k, []).append(n)
    lo_off, o = {}, 0
    for k, (mk, lk, pk) in enumerate(irreps_out):
        lo_off[k] = o
        o += mk * mk
    out, lo = [], 0
    for k in sorted(by_k, key=lambda k: ((irreps_out[k][1], irreps_out[k][2]), k)):
        mk = irreps_out[k][0]
        fan = mk * len(by_k[k])
        out.append((k, lo, fan, lo_off[k], mk))
        lo += fan * mk
    return out


def _tp_superpaths(nsrc: int, in_layout: PlanarLayout, irreps_sh: Irreps, irreps_out: Irreps, tp_weight, w3: np.ndarray,
                   lin_scale_w: np.ndarray, lin_out_w: Optional[np.ndarray], uvu: bool):
    """The algebra of ONE reference tensor-product branch in the edge-aligned frame, as "super-paths" (input irrep i, output irrep k):
    all e3nn paths (i, l_sh, k) stacked along `rows` (row = (path, mid channel w)).  Yields dicts with
        W  [nrows, mul_i * nsrc]  TP weights x path normalisation        (mid[row] = sum_u W[row, u] x_i[u])
        ch [nrows]                column of the last radial layer w3      (s[row]   = sum_h w3[h, ch[row]] h2[h])
        cf [nrows, 2 mm + 1]      aligned-frame CG coefficient per column (mm = min(l_i, l_k))
        L  [nrows, mul_k]         LinearScaleWithWeights.linear_out (x trailing o3.Linear) rows
        par                       1: column c reads input component l_i + mm - c (reversed), 0: l_i - mm + c
    and the flop count of the super-path per edge.  out[k][w'', c] += sum_rows L[row, w''] cf[row, c] s[row] mid[row, c]."""
    irr_in = Irreps([(m * nsrc, l, p) for m, l, p in in_layout.irreps])
    ins = tp_instructions(irr_in, irreps_sh, irreps_out)
    # flat TP weight offsets follow the instruction (slot) order; radial channels follow the sorted mid layout
    woff, choff = [], []
    wo = co = 0
    for (i, j, k, slot) in ins:
        woff.append(wo)
        choff.append(co)
        wo += 0 if uvu else irr_in[i][0] * irreps_out[k][0]
        co += irr_in[i][0] if uvu else irreps_out[k][0]
    assert wo == (0 if tp_weight is None else tp_weight.size), (wo, tp_weight.size)
    assert co == w3.shape[1], (co, w3.shape)
    # Linear(mid.simplify() -> irreps_out): simplified mid has one entry per distinct out irrep, in sorted order
    irs = [(l, p) for _, l, p in irreps_out]
    assert len(set(irs)) == len(irs), "duplicate irreps in the TP target are not supported by the planner"
    by_k: Dict[int, List[int]] = {}
    for n, (i, j, k, slot) in enumerate(ins):
        by_k.setdefault(k, []).append(n)
    # weight offsets of the Linear blocks: paths ordered by (i_in over sorted simplified mid, i_out)
    order = sorted(by_k, key=lambda k: ((irreps_out[k][1], irreps_out[k][2]), k))
    lin_off, lo = {}, 0
    for k in order:
        fan = sum((irr_in[ins[n][0]][0] if uvu else irreps_out[k][0]) for n in by_k[k])
        lin_off[k] = (lo, fan)
        lo += fan * irreps_out[k][0]
    assert lo == lin_scale_w.size, (lo, lin_scale_w.size)
    lo_off, o = {}, 0
    for k, (mk, lk, pk) in enumerate(irreps_out):                       # o3.Linear(out->out): one path per irrep
        lo_off[k] = o
        o += mk * mk
    if lin_out_w is not None:
        assert o == lin_out_w.size
    for k in order:
        mk, lk, pk = irreps_out[k]
        off, fan = lin_off[k]
        L = lin_scale_w[off:off + fan * mk].reshape(fan, mk).astype(np.float64) / math.sqrt(fan)
        if lin_out_w is not None:
            Lo = lin_out_w[lo_off[k]:lo_off[k] + mk * mk].reshape(mk, mk).astype(np.float64) / math.sqrt(mk)
            L = L @ Lo
        ch0 = choff[by_k[k][0]]
        # group the paths into k by input irrep i  (super-path (i,k): all l_sh stacked along rows)
        by_i: Dict[int, List[int]] = {}
        for n in by_k[k]:
            by_i.setdefault(ins[n][0], []).append(n)
        for i, plist in by_i.items():
            mi2, li, pi = irr_in[i]
            mm = min(li, lk)
            nc = 2 * mm + 1
            par = None
            rows_W, rows_ch, rows_cf, rows_L = [], [], [], []
            rows_meta = []                                     # per row: (instruction n, mid channel w, path normalisation, row of the k-block of L)
            flops = 0.0
            for n in plist:
                _, j, _, _ = ins[n]
                lj = irreps_sh[j][1]
                src_c, coef_c = so3.aligned_path(li, lj, lk)
                this_par = (li + lj + lk) % 2
                assert par is None or par == this_par
                par = this_par
                if uvu:                                        # unweighted uvu: coefficient sqrt(2 l_k + 1), channels pass through
                    mmid = mi2
                    W = np.eye(mi2) * math.sqrt(2 * lk + 1)
                else:
                    mmid = mk
                    cpath = math.sqrt((2 * lk + 1) / (mi2 * irreps_sh[j][0]))
                    W = tp_weight[woff[n]:woff[n] + mi2 * mk].reshape(mi2, mk).astype(np.float64) * cpath
                cf = np.array([coef_c[lk + m] for m in range(-mm, mm + 1)])
                l0 = choff[n] - ch0                            # the path's mmid rows, all at once (this runs on every weight repack)
                rows_W.append(W.T)
                rows_ch.append(choff[n] + np.arange(mmid))
                rows_cf.append(np.broadcast_to(cf, (mmid, nc)))
                rows_L.append(L[l0:l0 + mmid])
                rows_meta += zip([n] * mmid, range(mmid), [0.0 if uvu else cpath] * mmid, range(l0, l0 + mmid))
                flops += (0.0 if uvu else 2.0 * mi2 * mk * nc) + 2.0 * mmid * nc      # + 2 H mmid + 2 mmid mk nc, added by the caller (H)
                flops += 2.0 * mmid * mk * nc
            ch = np.concatenate(rows_ch)
            yield dict(i=i, k=k, mi=mi2 // nsrc, li=li, mk=mk, lk=lk, mm=mm, par=par, W=np.concatenate(rows_W), ch=ch,
                       cf=np.concatenate(rows_cf), L=np.concatenate(rows_L), flops=flops, nmid=len(ch), meta=rows_meta,
                       woff={n: woff[n] for n in plist}, lin=lin_off[k], lo_off=lo_off[k], pk=pk, pi=pi)


def add_tp_items(prog: Program, seg_of_k: Dict[int, int], in_layout: PlanarLayout, nsrc: int, srcs: Sequence[int],
                 irreps_sh: Irreps, irreps_out: Irreps, tp_weight: np.ndarray, w3: np.ndarray, lin_scale_w: np.ndarray,
                 lin_out_w: Optional[np.ndarray], mlp: int, uvu: bool = False, merge_groups: Sequence[Sequence[int]] = (), zero_inputs: Sequence[int] = (),
                 dead_out: Sequence[int] = ()):
    """Items of ONE reference tensor-product branch (node or edge) of a MessagePackBlock / embedding TP.
    dead_out: output irreps (indices into irreps_out) nobody reads where this block runs -- their super-paths are dropped and their tiles are written as
    zeros (see build_message_pack_program); a merge group must not contain one (choose_merge_groups(dead_out=...)).
    zero_inputs: input irreps (indices into in_layout.irreps) whose rows are STRUCTURALLY zero for this block -- every super-path that reads one of
    them contributes exactly nothing and is dropped (see build_message_pack_program).
    merge_groups: lists of output irreps k whose super-paths from one input irrep are stacked into ONE item (see Program.vsegs):
    the rows of a 16-row MFMA tile are then filled by several small output irreps instead of one (4x5o alone uses 12 of 16 rows of
    GEMM1 / the radial scale and 4 of 16 rows of GEMM2's output).  Only super-paths with l_i <= min l_k of the group are stacked (same
    column count 2 l_i + 1); the members must share the parity class (l_k + [p_k odd]) mod 2 so that `par` agrees.

    in_layout : planar layout of ONE source row (irreps of the un-doubled features); nsrc = 2 for the node branch
                (reference input = (2 mul) x ir with the first mul channels from src, the rest from dst: attention_utils.py:85-119).
    tp_weight : flat o3.TensorProduct.weight;  w3: last radial layer [H, n_chan] already divided by sqrt(H);
    lin_scale_w: flat LinearScaleWithWeights.linear_out.weight;  lin_out_w: flat trailing o3.Linear(out->out) or None.
    uvu       : lite_mode product (tensor_products.py:81-84,127-130): no TP weights, mid multiplicity = input multiplicity.
    """
    H = prog.hidden
    group_of = {k: gi for gi, G in enumerate(merge_groups) for k in G}
    lmin = [min(irreps_out[k][1] for k in G) for G in merge_groups]
    vid_of: Dict[int, int] = {}
    for gi, G in enumerate(merge_groups):                      # one virtual segment per group (shared by the branches of a program)
        members = [seg_of_k[k] for k in G]
        assert all(len(prog.seg_chunks[k]) == 1 for k in G)
        if members in prog.vsegs:
            vid_of[gi] = prog.vsegs.index(members)
        else:
            vid_of[gi] = len(prog.vsegs)
            prog.vsegs.append(members)
        for sg in members:
            prog.seg_key[sg] = members[0]
    stacked: Dict[Tuple[int, int], List[dict]] = {}
    plain: List[dict] = []
    zero_inputs = set(int(i) for i in zero_inputs)
    dead_out = set(int(k) for k in dead_out)
    assert not (dead_out & set(group_of)), "a merge group holds a dead output irrep"
    for sp in _tp_superpaths(nsrc, in_layout, irreps_sh, irreps_out, tp_weight, w3, lin_scale_w, lin_out_w, uvu):
        if sp["i"] in zero_inputs or sp["k"] in dead_out:
            continue
        gi = group_of.get(sp["k"])
        if gi is not None and sp["li"] <= lmin[gi]:
            stacked.setdefault((sp["i"], gi), []).append(sp)
        else:
            plain.append(sp)
    units = [(sp, seg_of_k[sp["k"]], None, sp["mk"], sp["L"]) for sp in plain]
    for (i, gi), sps in stacked.items():
        G = list(merge_groups[gi])
        voff, o = {}, 0
        for k in G:
            voff[k] = o
            o += irreps_out[k][0]
        sps = sorted(sps, key=lambda sp: G.index(sp["k"]))
        assert len({sp["par"] for sp in sps}) == 1 and len({sp["mm"] for sp in sps}) == 1, "merge group members must share the parity class"
        Lv = np.zeros((sum(sp["nmid"] for sp in sps), o))
        r = 0
        for sp in sps:
            Lv[r:r + sp["nmid"], voff[sp["k"]]:voff[sp["k"]] + sp["mk"]] = sp["L"]
            r += sp["nmid"]
        cat = dict(sps[0], W=np.concatenate([sp["W"] for sp in sps]), ch=np.concatenate([sp["ch"] for sp in sps]),
                   cf=np.concatenate([sp["cf"] for sp in sps]), flops=sum(sp["flops"] for sp in sps), nmid=Lv.shape[0])
        units.append((cat, seg_of_k[G[0]], vid_of[gi], o, Lv))
    for sp, seg, vid, mk, rows_L in units:
        i, mi, li, lk, mm, par = sp["i"], sp["mi"], sp["li"], sp["lk"], sp["mm"], sp["par"]
        if vid is None and mk > seg_rows_cap(lk):
            raise NotImplementedError(f"tensor-product target {mk}x(l={lk}) is wider than the {seg_rows_cap(lk)} channels one LDS tile holds")
        nc = 2 * mm + 1
        rows_W, rows_ch, rows_cf = sp["W"], sp["ch"], sp["cf"]
        prog.flops_per_row += sp["flops"] + 2.0 * H * sp["nmid"]
        nrows = len(rows_ch)
        chunk = rtm_max(nc) * 16
        ksteps = in_layout.mulp[i] // 4
        rto = prog.segs[seg][2] if vid is None else ceil_div(mk, 16)
        for r0 in range(0, nrows, chunk):
            r1 = min(nrows, r0 + chunk)
            n = r1 - r0
            rtm = ceil_div(n, 16)
            # physical row of logical row rho: within each 16-row tile the (g, r) index of the C fragment is transposed so
            # that GEMM2's K-step (rt, r) -- which reads rows {16 rt + 4 g + r : g} -- holds logical rows 16 rt + 4 r + g:
            # padding rows fill whole trailing K-steps and only ceil(n / 4) of the 4 rtm K-steps are issued (item[18])
            rho = np.arange(n)
            phys = 16 * (rho // 16) + 4 * (rho % 4) + (rho % 16) // 4
            R = rtm * 16
            a1 = []
            x4 = use_x4(in_layout.mulp[i], nc)
            for s_ in range(nsrc):
                Wk = np.zeros((mi, R))
                Wk[:, phys] = rows_W[r0:r1, s_ * mi:(s_ + 1) * mi].T             # [u, physical row]
                a1.append(_frag_A(Wk, ksteps, rtm, x4))
            a1_off = prog.add_weights(np.stack(a1))
            w3p = np.zeros((w3.shape[0], R))
            w3p[:, phys] = w3[:, rows_ch[r0:r1]]
            w3_off = _add_w3(prog, w3p, rtm)
            cfp = np.zeros((R, nc))
            cfp[phys] = rows_cf[r0:r1]
            cf_off = prog.add_weights(_cf_block(cfp, rtm, nc))       # [rt][c][g][r]
            Lp = np.zeros((R, rto * 16))
            Lp[phys, :mk] = rows_L[r0:r1]
            # A2[rt'][rt][lane][r]: L'[row = 16 rt + 4 (lane>>4) + r][w'' = 16 rt' + (lane&15)]
            a2 = Lp.reshape(rtm, 4, 4, rto, 16).transpose(3, 0, 1, 4, 2).reshape(rto, rtm, 64, 4)
            a2_off = prog.add_weights(a2)
            _add_item(prog, seg, IT_TP, list(srcs), in_layout.off[i], in_layout.mulp[i], li, mm, par, ksteps, rtm, mlp,
                      a1_off, w3_off, cf_off, a2_off, n, row_off=0 if vid is None else vid + 1, nk2=ceil_div(n, 4), rto=rto)


def add_tp_adjoint_items(prog: Program, in_layout: PlanarLayout, nsrc: int, src_g: int, gout_layout: PlanarLayout, irreps_sh: Irreps,
                         irreps_out: Irreps, tp_weight: np.ndarray, w3: np.ndarray, lin_scale_w: np.ndarray, lin_out_w: Optional[np.ndarray],
                         mlp: int, target_base: int, skip_inputs: Sequence[int] = ()):
    """DATA-GRADIENT items of one tensor-product branch: the adjoint of add_tp_items with respect to the branch's input rows, on the
    SAME kernels.  With out[k] = sum_rows L^T (cf s (W x_i)) the gradient is  g_x[i] = sum_rows W^T (cf' s (L g_out[k]))  -- the same
    item shape with the roles of the two weight matrices swapped: GEMM1 contracts the staged g_out block of irrep k (source slot
    `src_g`, layout `gout_layout`) with L, the radial scale and the CG coefficient are those of the forward item, GEMM2 applies W^T
    and accumulates into the tile of the program's output irrep `target_base + i` = the (nsrc * mul_i) x l_i block of the input
    gradient (sender channels first, then receiver: the reference's doubled input).  Column bookkeeping: the forward reads input
    component l_i - mm + c (par = 0) or l_i + mm - c (par = 1) for output column l_k - mm + c; the adjoint reads g_out component
    l_k - mm + c' resp. l_k + mm - c' for its output column l_i - mm + c', i.e. the same `neg` flag with c' = c resp. 2 mm - c.
    skip_inputs: input irreps whose gradient nobody reads (structurally zero inputs of a first layer: their producers only have the other irreps) -- the
    items that would compute it are not emitted, those blocks of the result are zeros."""
    H = prog.hidden
    skip_inputs = set(int(i) for i in skip_inputs)
    for sp in _tp_superpaths(nsrc, in_layout, irreps_sh, irreps_out, tp_weight, w3, lin_scale_w, lin_out_w, False):
        if sp["i"] in skip_inputs:
            continue
        i, k, mi, li, mk, lk, mm, par = sp["i"], sp["k"], sp["mi"], sp["li"], sp["mk"], sp["lk"], sp["mm"], sp["par"]
        nc = 2 * mm + 1
        rows_W, rows_ch, rows_L = sp["W"], sp["ch"], sp["L"]
        rows_cf = sp["cf"][:, ::-1] if par else sp["cf"]
        prog.flops_per_row += sp["flops"] + 2.0 * H * sp["nmid"]
        nrows = len(rows_ch)
        chunk = rtm_max(nc) * 16
        ksteps = gout_layout.mulp[k] // 4
        x4 = use_x4(gout_layout.mulp[k], nc)
        for seg, c0, c1 in prog.seg_chunks[target_base + i]:  # column chunks of the (nsrc * mul_i) target channels
            rto = prog.segs[seg][2]
            for r0 in range(0, nrows, chunk):
                r1 = min(nrows, r0 + chunk)
                n = r1 - r0
                rtm = ceil_div(n, 16)
                rho = np.arange(n)
                phys = 16 * (rho // 16) + 4 * (rho % 4) + (rho % 16) // 4        # see add_tp_items
                R = rtm * 16
                Lk = np.zeros((mk, R))
                Lk[:, phys] = rows_L[r0:r1].T                                     # [w'' (K of GEMM1), physical row]
                a1_off = prog.add_weights(_frag_A(Lk, ksteps, rtm, x4)[None])
                w3p = np.zeros((w3.shape[0], R))
                w3p[:, phys] = w3[:, rows_ch[r0:r1]]
                w3_off = _add_w3(prog, w3p, rtm)
                cfp = np.zeros((R, nc))
                cfp[phys] = rows_cf[r0:r1]
                cf_off = prog.add_weights(_cf_block(cfp, rtm, nc))
                Wp = np.zeros((R, rto * 16))
                Wp[phys, :c1 - c0] = rows_W[r0:r1, c0:c1]                          # [physical row, target channel u]
                a2 = Wp.reshape(rtm, 4, 4, rto, 16).transpose(3, 0, 1, 4, 2).reshape(rto, rtm, 64, 4)
                a2_off = prog.add_weights(a2)
                _add_item(prog, seg, IT_TP, [src_g], gout_layout.off[k], gout_layout.mulp[k], lk, mm, par, ksteps, rtm, mlp,
                          a1_off, w3_off, cf_off, a2_off, n, nk2=ceil_div(n, 4))


def add_linear_items(prog: Program, seg_of_k: Dict[int, int], in_layout: PlanarLayout, src: int, irreps_out: Irreps,
                     weight: np.ndarray, extra_scale: float = 1.0, zero_inputs: Sequence[int] = (), dead_out: Sequence[int] = ()):
    """Items of one o3.Linear(irreps_in -> irreps_out) (e3nn: paths ordered by (i_in, i_out), 1/sqrt(fan_in)); zero_inputs: structurally zero input
    irreps, dead_out: output irreps nobody reads -- their paths are dropped (the weights are still walked: the flat layout is the reference's)."""
    zero_inputs = set(int(i) for i in zero_inputs)
    dead_out = set(int(k) for k in dead_out)
    irr_in = in_layout.irreps
    paths = [(i, k) for i, (_, li, pi) in enumerate(irr_in) for k, (_, lk, pk) in enumerate(irreps_out) if (li, pi) == (lk, pk)]
    fan = {}
    for i, k in paths:
        fan[k] = fan.get(k, 0) + irr_in[i][0]
    off = 0
    for i, k in paths:
        mi, li, _ = irr_in[i]
        mk = irreps_out[k][0]
        Wfull = weight[off:off + mi * mk].reshape(mi, mk).astype(np.float64) * (extra_scale / math.sqrt(fan[k]))
        off += mi * mk
        if i in zero_inputs or k in dead_out:
            continue
        ksteps = in_layout.mulp[i] // 4
        # rows chunked like TP items so that the per-wave register budget is the same
        nc = 2 * li + 1
        chunk = rtm_max(nc) * 16
        for seg, c0, c1 in prog.seg_chunks[k]:
            W = Wfull[:, c0:c1]
            for r0 in range(0, c1 - c0, chunk):
                r1 = min(c1 - c0, r0 + chunk)
                rtm = ceil_div(r1 - r0, 16)
                a1_off = prog.add_weights(_frag_A(W[:, r0:r1], ksteps, rtm, use_x4(in_layout.mulp[i], nc))[None])
                _add_item(prog, seg, IT_LIN, [src], in_layout.off[i], in_layout.mulp[i], li, li, 0, ksteps, rtm, 0,
                          a1_off, 0, 0, 0, r1 - r0, row_off=r0)
        prog.flops_per_row += 2.0 * mi * mk * nc
    assert off == weight.size, (off, weight.size)


MAX_SEG_ROWS = 64        # output channels per segment (bounds the LDS tile); wider irreps are split column-wise


def seg_rows_cap(l: int) -> int:
    """output channels of one segment of an l-irrep: the wave-private LDS tile [rows + 1][(2l+1) 16 + 4] of the segment-stationary kernel
    has to stay below ~28 KB (4 waves x (tile + 11 KB operand ring) <= 160 KB): 64 rows up to l = 2, 48 / 32 / 32 / 32 / 16 for l = 3..7
    (the su2 head of an f-shell basis groups > 64 multiplicity-1 outputs per high-l irrep)."""
    return max(16, min(MAX_SEG_ROWS, ((7000 // ((2 * l + 1) * 16 + 4)) - 1) // 16 * 16))


def new_program(irreps_out, hidden=0, flags_of=lambda k, ir: 0) -> Tuple[Program, Dict[int, int]]:
    """seg_of_k[k] = segment id of output irrep k (first chunk); prog.seg_chunks[k] = [(segment, c0, c1), ...]."""
    lay = PlanarLayout(irreps_out)
    prog = Program(out_layout=lay, hidden=hidden)
    seg_of_k = {}
    prog.seg_chunks = {}
    for k, (mk, lk, pk) in enumerate(lay.irreps):
        chunks = []
        cap = seg_rows_cap(lk)
        for c0 in range(0, mk, cap):
            c1 = min(mk, c0 + cap)
            npad = lay.mulp[k] - mk if c1 == mk else 0         # channel-padding slots the last chunk zero-fills (flags bits 8..)
            sid = _add_segment(prog, lk, c1 - c0, k, flags_of(k, (mk, lk, pk)) | (npad << 8))
            prog.segs[sid][3] += c0                            # channel offset inside the planar block
            chunks.append((sid, c0, c1))
        seg_of_k[k] = chunks[0][0]
        prog.seg_chunks[k] = chunks
    return prog, seg_of_k

"""Small device tables: streaming o3.Linear (csrc/linear.hip), gate / norm activation, rotation channels, row programs (csrc/rowprog.hip), attention heads,
the read-out's CG merge maps (non-SOC, SOC / su2) and the symmetric-contraction tables of the CorrProductBlock."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from .. import so3
from ..so3 import Irreps
from ._blas import single_thread_blas
from .layout import ACT_NONE, ACT_SSP, ACT_TANH, IT_LIN, PlanarLayout, ceil_div, rtm_max
from .program import _WEIGHT_DTYPE, _add_item, _frag_A, new_program, use_x4

# ---- streaming block-Linear (csrc/linear.hip): o3.Linear on planar rows as one HBM-bound pass --------------------------------
LIN_CHUNK = 64           # output channels per unit (4 MFMA row tiles of accumulators per wave)
LIN_UNIT_I32, LIN_PATH_I32, LIN_GROUP_I32 = 8, 4, 4


@dataclass
class LinearTables:
    """tables of hg_linear_planar.  A "pair-row" is one (row, component a) of an irrep block: `mulp` contiguous floats.
    groups int32[ngroup][4] = {unit_begin, nchunks, nco = 2 l + 1, 0}: one output irrep block = its channel chunks (units);
    units  int32[nunit][8]  = {out_off, out_mulp, rtm, store_channels (multiple of 4), path_begin, path_end, 0, 0};
    paths  int32[npath][4]  = {in_off, in_mulp, ngrp, w_off}: A fragments [ngrp][rtm][64][4] of the (normalised) weight block
                              W^T[out channel][in channel], K permuted for float4 B loads (_frag_A(..., x4=True))."""
    groups: np.ndarray
    units: np.ndarray
    paths: np.ndarray
    weights: np.ndarray
    items: np.ndarray            # int32[nitems][2] = {unit, component a}: the wave units of one block of rows, heaviest output block first
    in_dim: int
    out_dim: int
    flops_per_row: float


@single_thread_blas
def linear_tables(mats: Dict[Tuple[int, int], np.ndarray], in_layout: PlanarLayout, out_layout: PlanarLayout, keep_zero_blocks: bool = False) -> LinearTables:
    """mats[(i, k)] = [mul_i, mul_k] weight block (normalisation folded in) from input irrep i to output irrep k of the two planar
    layouts (same l, p).  Every output block is written in full (blocks without a path: zeros), padding channels included.
    keep_zero_blocks: the table structure does not depend on the weight VALUES (device-side refresh after an optimiser step, nn.E3Linear)."""
    groups, units, paths, chunks = [], [], [], []
    woff, flops = 0, 0.0
    order = sorted(range(len(out_layout.irreps)), key=lambda k: -sum(m.shape[0] for (i, kk), m in mats.items() if kk == k) * (2 * out_layout.irreps[k][1] + 1))
    for k in order:
        mk, lk, pk = out_layout.irreps[k]
        mulp = out_layout.mulp[k]
        ins = sorted(i for (i, kk) in mats if kk == k)
        groups.append([len(units), ceil_div(mulp, LIN_CHUNK), 2 * lk + 1, 0])
        for c0 in range(0, mulp, LIN_CHUNK):
            c1 = min(mulp, c0 + LIN_CHUNK)
            rtm = ceil_div(c1 - c0, 16)
            pb = len(paths)
            for i in ins:
                M = np.asarray(mats[(i, k)], dtype=np.float64)
                mi = in_layout.irreps[i][0]
                assert M.shape == (mi, mk) and in_layout.irreps[i][1:] == (lk, pk)
                blk = M[:, c0:min(c1, mk)]
                if blk.shape[1] == 0 or not (keep_zero_blocks or np.any(blk)):
                    continue
                ngrp = ceil_div(in_layout.mulp[i], 16)
                frag = _frag_A(blk, 4 * ngrp, rtm, True).astype(_WEIGHT_DTYPE[0]).reshape(-1)
                paths.append([in_layout.off[i], in_layout.mulp[i], ngrp, woff])
                chunks.append(frag)
                woff += frag.size
            units.append([out_layout.off[k] + c0, mulp, rtm, c1 - c0, pb, len(paths), 0, 0])
        flops += sum(2.0 * mats[(i, k)].shape[0] * mk * (2 * lk + 1) for i in ins)
    items = [[u0 + c, a] for u0, nch, nco, _ in groups for a in range(nco) for c in range(nch)]      # chunks of one (block, a) adjacent: shared input
    return LinearTables(np.asarray(groups, np.int32).reshape(-1, LIN_GROUP_I32), np.asarray(units, np.int32).reshape(-1, LIN_UNIT_I32),
                        np.asarray(paths, np.int32).reshape(-1, LIN_PATH_I32),
                        np.concatenate(chunks) if chunks else np.zeros(4, _WEIGHT_DTYPE[0]), np.asarray(items, np.int32).reshape(-1, 2),
                        in_layout.dim, out_layout.dim, flops)


def o3_linear_mats(weight: np.ndarray, irreps_in, irreps_out) -> Dict[Tuple[int, int], np.ndarray]:
    """weight blocks of e3nn's o3.Linear(irreps_in -> irreps_out): paths ordered (i_in, i_out), each (mul_in, mul_out) row-major,
    normalised by 1 / sqrt(fan_in of the output irrep)."""
    irreps_in, irreps_out = Irreps(irreps_in), Irreps(irreps_out)
    pth = [(i, k) for i, (_, li, pi) in enumerate(irreps_in) for k, (_, lk, pk) in enumerate(irreps_out) if (li, pi) == (lk, pk)]
    fan: Dict[int, int] = {}
    for i, k in pth:
        fan[k] = fan.get(k, 0) + irreps_in[i][0]
    mats, off = {}, 0
    weight = np.asarray(weight, dtype=np.float64).reshape(-1)
    for i, k in pth:
        mi, mk = irreps_in[i][0], irreps_out[k][0]
        mats[(i, k)] = weight[off:off + mi * mk].reshape(mi, mk) / math.sqrt(fan[k])
        off += mi * mk
    assert off == weight.size, (off, weight.size)
    return mats


@single_thread_blas
def build_linear_tables(weight: np.ndarray, irreps_in, irreps_out, keep_zero_blocks: bool = False) -> LinearTables:
    return linear_tables(o3_linear_mats(weight, irreps_in, irreps_out), PlanarLayout(irreps_in), PlanarLayout(irreps_out), keep_zero_blocks)


@single_thread_blas
def build_linear_adjoint_tables(weight: np.ndarray, irreps_in, irreps_out, keep_zero_blocks: bool = False) -> LinearTables:
    """data gradient of o3.Linear(irreps_in -> irreps_out) as tables of the same streaming kernel: g_x[i] = sum_k W_ik^T g_y[k] with the
    forward's normalised blocks transposed."""
    mats = {(k, i): M.T for (i, k), M in o3_linear_mats(weight, irreps_in, irreps_out).items()}
    return linear_tables(mats, PlanarLayout(irreps_out), PlanarLayout(irreps_in), keep_zero_blocks)


def wigner_jtab(lmax) -> np.ndarray:
    Js, sg = [], []
    for l in range(lmax + 1):
        J, s = so3.wigner_tables(l)
        Js.append(J.reshape(-1))
        sg.append(s)
    return np.concatenate(Js + sg).astype(np.float32)


def rotate_table(layout: PlanarLayout) -> np.ndarray:
    """int32[ngroups][4] = {l, planar offset of (component 0, first channel of the group), mulp, valid channels (1..4)}: one entry per
    group of 4 channel slots, sorted by l (stable) so that the wavefronts of hg_rotate_gather run a single <L> code path."""
    rows = []
    for (mul, l, p), off, mp in zip(layout.irreps, layout.off, layout.mulp):
        if l > 7:
            raise NotImplementedError(f"feature irreps with l = {l} > 7 have no rotation kernel instantiation")
        for u in range(0, mp, 4):
            rows.append((l, off + u, mp, max(0, min(4, mul - u))))
    rows.sort(key=lambda r: r[0])
    return np.asarray(rows, dtype=np.int32).reshape(-1, 4)


def gate_tables(feature_irreps):
    """Layouts + element table of the reference ResidualBlock's e3nn Gate (interaction_blocks.py:311-323; irreps2gate
    utils/irreps_utils.py:33-65; e3nn Gate = _Sortcut(sorted+simplified input) -> Activation / ElementwiseTensorProduct).
    Returns (irreps_gate_in, irreps_gate_out, table int32[Dout_planar][4])."""
    feats = Irreps(feature_irreps)
    scalars = Irreps([(m, l, p) for m, l, p in feats if l == 0]).simplify()
    gated = Irreps([(m, l, p) for m, l, p in feats if l != 0]).simplify()
    gates = Irreps([(m, 0, 1) for m, _, _ in gated]).simplify()
    entries = [("s", i, it) for i, it in enumerate(scalars)] + [("g", i, it) for i, it in enumerate(gates)] + \
              [("d", i, it) for i, it in enumerate(gated)]
    order = sorted(range(len(entries)), key=lambda i: ((entries[i][2][1], entries[i][2][2]), i))
    merged, where = [], {}                      # where[(grp, idx)] = (merged entry, channel offset)
    for i in order:
        grp, idx, (m, l, p) = entries[i]
        if merged and merged[-1][1:] == (l, p):
            where[(grp, idx)] = (len(merged) - 1, merged[-1][0])
            merged[-1] = (merged[-1][0] + m, l, p)
        else:
            where[(grp, idx)] = (len(merged), 0)
            merged.append((m, l, p))
    irr_in = Irreps(merged)
    lay_in = PlanarLayout(irr_in)
    # scalar activations: even scalars -> ssp, odd -> tanh (odd act keeps parity); gates (0e) -> ssp
    out_items = [(m, 0, p) for m, _, p in scalars] + list(gated.items)
    irr_out = Irreps(out_items)
    lay_out = PlanarLayout(irr_out)
    tab = np.full((lay_out.dim, 4), -1, dtype=np.int32)
    for i, (m, _, p) in enumerate(scalars):
        me, u0 = where[("s", i)]
        act = ACT_SSP if p == 1 else ACT_TANH
        for u in range(m):
            tab[lay_out.off[i] + u] = (lay_in.off[me] + u0 + u, act, -1, 0)
    gate_pos = []                               # planar input index of every gate channel, in gate order
    for i, (m, _, _) in enumerate(gates):
        me, u0 = where[("g", i)]
        gate_pos += [lay_in.off[me] + u0 + u for u in range(m)]
    gc = 0
    for i, (m, l, p) in enumerate(gated):
        me, u0 = where[("d", i)]
        oi = len(scalars) + i
        for u in range(m):
            for a in range(2 * l + 1):
                tab[lay_out.off[oi] + a * lay_out.mulp[oi] + u] = (lay_in.off[me] + a * lay_in.mulp[me] + u0 + u, ACT_NONE, gate_pos[gc + u], ACT_SSP)
        gc += m
    return irr_in, irr_out, tab


def norm_act_table(irreps) -> np.ndarray:
    """int32[nchan][2] = {offset of the irrep copy's first component in the planar row, component stride | components << 16} of csrc/aux_kernels.hip:
    norm_act_kernel (e3nn NormActivation: one norm per irrep COPY)"""
    lay = PlanarLayout(irreps)
    out = []
    for i, (mul, l, _) in enumerate(lay.irreps):
        assert lay.mulp[i] < (1 << 16)
        for u in range(mul):
            out.append([lay.off[i] + u, lay.mulp[i] | ((2 * l + 1) << 16)])
    return np.asarray(out, np.int32).reshape(-1, 2)


def gate_tables_compact(tab: np.ndarray):
    """tables of hg_gate from gate_tables' [Dout][4] = {src, act, gate, gate act}: the distinct (input, activation) pairs are listed once
    (act_tab) and the outputs refer to them by slot, so a gate channel's activation is evaluated once per row instead of once per
    component of the irrep it gates.  Returns (act_tab int32[nact][2], out_tab int32[Dout][2])."""
    slots: Dict[Tuple[int, int], int] = {}

    def slot(idx, act):
        return slots.setdefault((int(idx), int(act)), len(slots))
    out = np.full((tab.shape[0], 2), -1, dtype=np.int32)
    for p, (src, act, gate, gact) in enumerate(tab):
        if src < 0:
            continue
        out[p, 0] = src if act == ACT_NONE else (0x40000000 | slot(src, act))
        if gate >= 0:
            out[p, 1] = slot(gate, gact)
    act_tab = np.zeros((max(1, len(slots)), 2), dtype=np.int32)
    for (idx, act), k in slots.items():
        act_tab[k] = (idx, act)
    return act_tab[:len(slots)] if slots else act_tab[:0], out


# ------------------------------------------------------------------------------------------------ row programs (csrc/rowprog.hip)
RP_NW = 16                  # waves of a workgroup (1024 threads on one tile of 16 rows, one workgroup per CU)
RP_ROWS = 16
RP_STAGE_I32 = 24
RP_UNIT_I32 = 12
RP_LINEAR, RP_GATE = 1, 2
RP_LDS_MAX = 160 * 1024


@dataclass
class RowProgram:
    """A chain of row-local stages run on 16 rows held in LDS (csrc/rowprog.hip): o3.Linear blocks as MFMA units reading one LDS buffer and
    writing the other (optionally accumulating onto what is there: the residual add), e3nn Gates in place.  HamLayer.forward
    (hamgnn_output.py:51-58) = Linear1 -> Gate -> Linear2 (+ x) -> linear_transform is one program: one read of the feature row, one write of
    the coefficient row, no intermediate row leaves the chip.
    stages int32[n][RP_STAGE_I32]: {type, src buffer, dst buffer, unit range of wave 0..RP_NW (RP_NW + 1 ints) | gate: act_tab offset, nact, out_tab
    offset, Dout at [3..6]; act_tab rows {input index, activation}: applied IN PLACE; out_tab rows {source index | -1, gate index | -1}};  units int32[n][RP_UNIT_I32]: {in_off, in_mulp, K-steps of 4, out_off (tile of 16 channels), out_mulp, components,
    valid float4 groups of the tile, weight offset, accumulate, the wave's next unit, 0...};  weights: A-operand fragments [ceil(steps / 4)][64][4] per unit
    (lane (out channel, kk) holds W[4 (4 G + q) + kk][channel], q = float4 component);  rs: LDS row strides of the two buffers (== 4 mod 64)."""
    stages: np.ndarray
    units: np.ndarray
    weights: np.ndarray
    act_tab: np.ndarray
    out_tab: np.ndarray
    din: int
    dout: int
    in_buf: int
    out_buf: int
    rs: Tuple[int, int]
    strip: int
    lds_bytes: int
    flops_per_row: float
    mfma_per_tile: int


@single_thread_blas
def build_row_program(specs, din: int) -> RowProgram:
    """specs: list of ("linear", mats {(i, k): [mul_i, mul_k]}, in_layout, out_layout, accumulate) | ("gate", table of plan.gate_tables, Din, Dout);
    the first stage reads buffer 0 (the staged input rows), every linear stage writes the other buffer, a gate works in place."""
    stages, units, wparts, acts, outs = [], [], [], [], []
    woff = 0
    cur, width = 0, [din, 0]
    flops, mfmas, strip = 0.0, 0, 0
    for spec in specs:
        if spec[0] == "gate":
            _, tab, gin, gout = spec
            act_tab, out_c = gate_tables_compact(np.asarray(tab))
            # in place: the activated scalars overwrite their inputs (every (input, activation) pair is distinct and no input carries two
            # activations), so the outputs look values up by INPUT index and no activation strip is needed
            assert len({int(i) for i, _ in act_tab}) == len(act_tab)
            out_tab = np.full_like(out_c, -1)
            for p_, (src, gate) in enumerate(out_c):
                if src >= 0:
                    out_tab[p_, 0] = act_tab[src & 0x3fffffff][0] if (src & 0x40000000) else src
                    out_tab[p_, 1] = act_tab[gate][0] if gate >= 0 else -1
            assert gin <= width[cur] or True
            rec = [RP_GATE, cur, cur, sum(len(a) for a in acts), len(act_tab), sum(len(o) for o in outs), int(gout)] + [0] * (RP_STAGE_I32 - 7)
            acts.append(act_tab.reshape(-1, 2))
            outs.append(out_tab.reshape(-1, 2))
            width[cur] = max(width[cur], int(gin), int(gout))
            if gout > 16 * 64:
                raise NotImplementedError("row program: gate rows wider than 1024 floats")
            stages.append(rec)
            continue
        _, mats, lin, lout, accumulate = spec
        dst = 1 - cur
        width[cur] = max(width[cur], lin.dim)
        width[dst] = max(width[dst], lout.dim)
        tiles = []                                             # (cost, [unit records]) per (output irrep, tile of 16 channels)
        for k, (mk, lk, pk) in enumerate(lout.irreps):
            ins = sorted(i for (i, kk) in mats if kk == k)
            ncomp, mulp = 2 * lk + 1, lout.mulp[k]
            for c0 in range(0, mulp, 16):
                recs, first = [], True
                for i in ins:
                    M = np.asarray(mats[(i, k)], dtype=np.float64)
                    blk = np.zeros((lin.mulp[i], 16))
                    w = M[:, c0:min(c0 + 16, mk)]
                    blk[:w.shape[0], :w.shape[1]] = w
                    if not np.any(blk):
                        continue
                    if lin.mulp[i] > 64:
                        raise NotImplementedError("row program: more than 64 channels per input irrep")
                    nsteps = lin.mulp[i] // 4
                    frag = _frag_A(blk, nsteps, 1, False).reshape(-1)
                    recs.append([lin.off[i], lin.mulp[i], nsteps, lout.off[k] + c0, mulp, ncomp, min(4, (mulp - c0) // 4), woff, 0 if (first and not accumulate) else 1, 0, 0, 0])
                    wparts.append(frag)
                    woff += frag.size
                    first = False
                    flops += 2.0 * M.shape[0] * w.shape[1] * ncomp
                    mfmas += nsteps * ncomp
                if not recs and not accumulate:                # an output block without a path: zeros (o3.Linear leaves it at zero)
                    recs.append([0, 0, 0, lout.off[k] + c0, mulp, ncomp, min(4, (mulp - c0) // 4), 0, 0, 0, 0, 0])
                if recs:
                    tiles.append((sum(r[2] for r in recs) * ncomp + 2 * ncomp, recs))
        tiles.sort(key=lambda t: -t[0])
        load = [0] * RP_NW
        mine = [[] for _ in range(RP_NW)]
        for cost, recs in tiles:                               # longest first onto the least loaded wave; the units of a tile stay with one wave, in order
            w_ = int(np.argmin(load))
            load[w_] += cost
            mine[w_] += recs
        begin = [len(units)]
        for w_ in range(RP_NW):
            units += mine[w_]
            begin.append(len(units))
        stages.append([RP_LINEAR, cur, dst] + begin + [0] * (RP_STAGE_I32 - 3 - len(begin)))
        cur = dst
    # every unit names its wave's NEXT unit (the following stage's first, and after the last stage the first unit again: the next tile), whose
    # weight fragments the kernel requests while this one computes
    for w_ in range(RP_NW):
        chain = [u for st in stages if st[0] == RP_LINEAR for u in range(st[3 + w_], st[4 + w_])]
        for a_, b_ in zip(chain, chain[1:] + chain[:1]):
            units[a_][9] = b_
    rs = tuple(int(w + ((4 - w) % 64)) if w else 4 for w in width)
    lds = 4 * (RP_ROWS * (rs[0] + rs[1]) + RP_NW * strip)
    if lds > RP_LDS_MAX:
        raise NotImplementedError("row program: the two row buffers do not fit the LDS")
    dout = width[cur] if stages[-1][0] == RP_GATE else specs[-1][3].dim
    cat = lambda l_: (np.concatenate(l_).astype(np.int32) if l_ else np.zeros((0, 2), np.int32))
    return RowProgram(np.asarray(stages, np.int32).reshape(-1, RP_STAGE_I32), np.asarray(units, np.int32).reshape(-1, RP_UNIT_I32),
                      np.concatenate(wparts).astype(np.float32) if wparts else np.zeros(4, np.float32), cat(acts), cat(outs), int(din), int(dout), 0, cur, rs, strip, lds,
                      flops, mfmas)


def attention_head_table(irreps, num_heads: int):
    """head of every planar column for AttentionAggregation (hamgnn/nn/attention.py:103-123, attention_utils.py:28-45): the reference
    views each (mul x ir) block as [heads, mul / heads * dim], i.e. head h = channels [h mul/H, (h+1) mul/H) of the block, all m.
    Returns (int32[Dp] head or -1 for padding columns, head dimension sum_k mul_k / H * (2 l_k + 1))."""
    irreps = Irreps(irreps)
    lay = PlanarLayout(irreps)
    if not 1 <= num_heads <= 8:
        raise NotImplementedError(f"num_heads = {num_heads}: the attention kernels hold 1..8 heads")
    tab = np.full(lay.dim, -1, dtype=np.int32)
    head_dim = 0
    for k, (mul, l, p) in enumerate(irreps):
        if mul % num_heads:
            raise ValueError(f"irreps multiplicity {mul} (l={l}) is not divisible by num_heads={num_heads} "
                             "(the reference's view(N, heads, -1) needs that, attention_utils.py:39-45)")
        per = mul // num_heads
        head_dim += per * (2 * l + 1)
        for a in range(2 * l + 1):
            o = lay.off[k] + a * lay.mulp[k]
            tab[o:o + mul] = np.arange(mul) // per
    return tab, head_dim


def ham_irreps(row: Irreps):
    """hamiltonian_irreps of the reference head (hamgnn_output.py:258-272): per (row shell, col shell) all L, parity (-1)^(li+lj)."""
    out = []
    for _, li, _ in row:
        for _, lj, _ in row:
            for L in range(abs(li - lj), li + lj + 1):
                out.append((1, L, (-1) ** (li + lj)))
    return Irreps(out)


@single_thread_blas
def ham_linear_mats(weight: np.ndarray, irreps_in, hirr: Irreps, keep=None):
    """o3.Linear(D -> hamiltonian_irreps) (HamLayer.linear_transform, hamgnn_output.py:49,56) regrouped by (L,p) so that the
    89..312 multiplicity-1 outputs become a handful of GEMM blocks.  Returns (normalised weight blocks {(i_in, group): [mul_in, n_group]},
    grouped irreps, slot -> (group, col)).
    keep: optional bool per output slot; slots not kept own weights (checkpoint layout) but are never computed (the su2 head
    reads only half of its 2 x 2 x required irreps, tensor_decomposition.py:545-551)."""
    irreps_in = Irreps(irreps_in)
    keep = [True] * len(hirr) if keep is None else list(keep)
    # an output irrep without a matching input irrep has no o3.Linear path: e3nn leaves it at zero.  Such slots (e.g. the l = 7 outputs
    # of the su2 head of f-shell bases fed by l <= 6 features) are never computed; the merge tables treat them as zero coefficients.
    have = {(l, p) for _, l, p in irreps_in}
    keep = [k and ((L, p) in have) for k, (_, L, p) in zip(keep, hirr)]
    groups, slot_pos = [], []
    key_to_g = {}
    for s, (_, L, p) in enumerate(hirr):
        if not keep[s]:
            slot_pos.append(None)
            continue
        if (L, p) not in key_to_g:
            key_to_g[(L, p)] = len(groups)
            groups.append([0, L, p])
        g = key_to_g[(L, p)]
        slot_pos.append((g, groups[g][0]))
        groups[g][0] += 1
    girr = Irreps([tuple(g) for g in groups])
    # e3nn flat weight order: for i_in, for i_out (matching ir): block (mul_in, 1)
    mats = {}
    fan = {}
    off = 0
    for i, (mi, li, pi) in enumerate(irreps_in):
        for s, (_, L, p) in enumerate(hirr):
            if (li, pi) == (L, p):
                if keep[s]:
                    g, col = slot_pos[s]
                    M = mats.setdefault((i, g), np.zeros((mi, girr[g][0])))
                    M[:, col] = weight[off:off + mi]
                    fan[s] = fan.get(s, 0) + mi
                off += mi
    assert off == weight.size, (off, weight.size)
    for (i, g) in list(mats):
        cols_fan = np.array([fan[s] for s in range(len(hirr)) if keep[s] and slot_pos[s][0] == g], dtype=np.float64)
        mats[(i, g)] = mats[(i, g)] / np.sqrt(cols_fan)[None, :]
    return mats, girr, slot_pos


@single_thread_blas
def build_ham_linear_program(weight: np.ndarray, irreps_in, hirr: Irreps, keep=None):
    """ham_linear_mats as a program of the segment-stationary kernel (HG_LINEAR_KERNEL=seg).  Returns (program, grouped irreps, slot->(group, col))."""
    irreps_in = Irreps(irreps_in)
    mats, girr, slot_pos = ham_linear_mats(weight, irreps_in, hirr, keep)
    prog, seg_of_k = new_program(girr, 0)
    in_layout = PlanarLayout(irreps_in)
    for (i, g), Mn in mats.items():
        mi, li, _ = irreps_in[i]
        mk = girr[g][0]
        nc = 2 * li + 1
        chunk = rtm_max(nc) * 16
        ksteps = in_layout.mulp[i] // 4
        for seg, c0, c1 in prog.seg_chunks[g]:
            for r0 in range(0, c1 - c0, chunk):
                r1 = min(c1 - c0, r0 + chunk)
                rtm = ceil_div(r1 - r0, 16)
                a1_off = prog.add_weights(_frag_A(Mn[:, c0 + r0:c0 + r1], ksteps, rtm, use_x4(in_layout.mulp[i], nc))[None])
                _add_item(prog, seg, IT_LIN, [0], in_layout.off[i], in_layout.mulp[i], li, li, 0, ksteps, rtm, 0, a1_off, 0, 0, 0, r1 - r0, row_off=r0)
        prog.flops_per_row += 2.0 * mi * mk * nc
    return prog.finalize(), girr, slot_pos


def su2_irreps(row: Irreps) -> Irreps:
    """One complex half of E3TensorDecomposition(spinful=True).required_irreps_out (hamgnn/nn/tensor_decomposition.py:40-88,
    463-486): per (row shell, col shell) the L list of l_i x l_j, then for every L the coupling with the spin vector
    L x 1 -> |L-1|..L+1; parity (-1)^(l_i+l_j) throughout."""
    out = []
    for _, li, _ in row:
        for _, lj, _ in row:
            p = (-1) ** (li + lj)
            Ls = range(abs(li - lj), li + lj + 1)
            out += [(1, L, p) for L in Ls]
            out += [(1, l2, p) for L in Ls for l2 in range(abs(L - 1), L + 2)]
    return Irreps(out)


def su2_merge_tables(row: Irreps, nao, index_change, minus_index, girr: Irreps, slot_pos):
    """slot table + CSR table of hg_ham_merge for the SOC/su2 head: E3TensorDecomposition.get_H (tensor_decomposition.py:
    553-603) + reorder_matrix (hamgnn_output.py:1056-1096) + the (2,2,nao,nao)->(2 nao, 2 nao) spin-block interleave
    (:3151-3152) as ONE real-linear map from the used network outputs (re: copy 0, im: copy 2 of the 4 x required irreps)
    to [real plane | imag plane], each (2 nao)^2.  slot_pos indexes the full 4-copy irreps list."""
    glay = PlanarLayout(girr)
    half = su2_irreps(row)
    S = len(half)
    R = sum(2 * L + 1 for _, L, _ in half)
    assert R == 4 * nao * nao
    slot_tab = np.zeros((2 * R, 4), dtype=np.int32)
    q = 0
    dead = np.zeros(2 * R, dtype=bool)               # coefficients of outputs the Linear has no path to (identically zero)
    for copy in (0, 2):
        for s, (_, L, p) in enumerate(half):
            pos = slot_pos[copy * S + s]
            for a in range(2 * L + 1):
                if pos is None:
                    slot_tab[q] = (0, 0, 0, 0)       # reads a finite value; every CSR entry pointing here is dropped below
                    dead[q] = True
                else:
                    slot_tab[q] = (L, a, glay.off[pos[0]] + pos[1], glay.mulp[pos[0]])
                q += 1
    s2 = math.sqrt(2.0)
    spin = np.array([[1, 0, 1, 0], [0, -1j, 0, 1], [0, 1j, 0, 1], [1, 0, -1, 0]], dtype=np.complex128) / s2
    entries = {}                                     # (j, A, B) pre-reorder -> (coef index array, complex values)
    off, r0 = 0, 0
    for _, li, _ in row:
        c0 = 0
        ni = 2 * li + 1
        for _, lj, _ in row:
            nj = 2 * lj + 1
            Ls = list(range(abs(li - lj), li + lj + 1))
            m = ni * nj
            wm = np.concatenate([so3.wigner_3j(li, lj, L) for L in Ls], axis=-1)          # [ni, nj, m]
            T = np.zeros((4, ni, nj, 4 * m), dtype=np.complex128)
            T[:, :, :, :m] = np.einsum("j,abm->jabm", spin[:, 0], wm)
            o2, mo = m, 0
            for L in Ls:
                Lp = list(range(abs(L - 1), L + 2))
                wsp = np.concatenate([so3.wigner_3j(L, 1, l2) for l2 in Lp], axis=-1)       # [2L+1, 3, d]
                d = wsp.shape[-1]
                T[:, :, :, o2:o2 + d] = np.einsum("jn,abM,Mnl->jabl", spin[:, 1:], wm[:, :, mo:mo + 2 * L + 1], wsp)
                o2 += d
                mo += 2 * L + 1
            assert o2 == 4 * m
            for j in range(4):
                for a in range(ni):
                    for b in range(nj):
                        v = T[j, a, b]
                        nz = np.nonzero(np.abs(v) > 1e-14)[0]
                        entries[(j, r0 + a, c0 + b)] = (off + nz, v[nz])
            off += 4 * m
            c0 += nj
        r0 += ni
    assert off == R
    ic = list(range(nao)) if index_change is None else list(index_change)
    sign = np.ones(nao)
    if minus_index is not None:
        sign[list(minus_index)] = -1
    n2 = 2 * nao
    ptr, idx, val = [0], [], []
    for plane in (0, 1):
        for Rr in range(n2):
            for Cc in range(n2):
                s1, r = divmod(Rr, nao)
                s2_, c = divmod(Cc, nao)
                k, v = entries[(2 * s1 + s2_, ic[r], ic[c])]
                v = v * (sign[r] * sign[c])
                re_c, im_c = (v.real, -v.imag) if plane == 0 else (v.imag, v.real)      # (A_r + i A_i)(x + i y)
                for kk, w in zip(k, re_c):
                    if abs(w) > 1e-14 and not dead[kk]:
                        idx.append(kk); val.append(w)
                for kk, w in zip(k, im_c):
                    if abs(w) > 1e-14 and not dead[R + kk]:
                        idx.append(R + kk); val.append(w)
                ptr.append(len(idx))
    return slot_tab, np.asarray(ptr, np.int32), np.asarray(idx, np.int32), np.asarray(val, np.float32)


def ham_merge_tables(row: Irreps, nao, index_change, minus_index, girr: Irreps, slot_pos):
    """slot table + CSR Clebsch-Gordan table of hg_ham_merge (merge_tensor_components hamgnn_output.py:851-891 followed by
    reorder_matrix :1056-1096 folded in)."""
    glay = PlanarLayout(girr)
    hirr = ham_irreps(row)
    slot_tab = np.zeros((nao * nao, 4), dtype=np.int32)
    coef_off = []
    dead = np.zeros(nao * nao, dtype=bool)           # coefficients of outputs the Linear has no path to (identically zero)
    q = 0
    for s, (_, L, p) in enumerate(hirr):
        pos = slot_pos[s]
        coef_off.append(q)
        for a in range(2 * L + 1):
            if pos is None:
                dead[q] = True                       # slot reads a finite value; its CSR entries are dropped below
            else:
                slot_tab[q] = (L, a, glay.off[pos[0]] + pos[1], glay.mulp[pos[0]])
            q += 1
    assert q == nao * nao
    entries = [[] for _ in range(nao * nao)]          # per merged (pre-reorder) element: list of (coef index, value)
    s = 0
    r0 = 0
    for _, li, _ in row:
        c0 = 0
        for _, lj, _ in row:
            for L in range(abs(li - lj), li + lj + 1):
                cg = math.sqrt(2 * L + 1) * so3.wigner_3j(li, lj, L)
                for a in range(2 * li + 1):
                    for b in range(2 * lj + 1):
                        for M in range(2 * L + 1):
                            if abs(cg[a, b, M]) > 1e-14 and not dead[coef_off[s] + M]:
                                entries[(r0 + a) * nao + (c0 + b)].append((coef_off[s] + M, cg[a, b, M]))
                s += 1
            c0 += 2 * lj + 1
        r0 += 2 * li + 1
    ic = list(range(nao)) if index_change is None else list(index_change)
    sign = np.ones(nao)
    if minus_index is not None:
        sign[list(minus_index)] = -1
    ptr, idx, val = [0], [], []
    for r in range(nao):
        for c in range(nao):
            sg = sign[r] * sign[c]
            for (ci, v) in entries[ic[r] * nao + ic[c]]:
                idx.append(ci)
                val.append(sg * v)
            ptr.append(len(idx))
    return slot_tab, np.asarray(ptr, np.int32), np.asarray(idx, np.int32), np.asarray(val, np.float32)


def ham_merge_adjoint_tables(slot_tab: np.ndarray, ptr: np.ndarray, idx: np.ndarray, val: np.ndarray, planar_dim: int):
    """Data gradient of hg_ham_merge from its own tables: with H = C (D^T y) (C = the CSR map, D^T = the per-irrep
    un-rotation of the planar coefficient rows y) the gradient is g_y = D (C^T g_H).  Returns
      slot_id int32[nout][4]  identity slots (the first launch reads g_H columns as they are, no rotation),
      (ptrT, idxT, valT)      C^T as CSR over the coefficients,
      scatter int32[planar_dim]  planar column -> coefficient index (or -1): hg_from_planar places C^T g_H into the planar rows,
    after which hg_rotate_gather (not transposed) applies D."""
    nout, ncoef = len(ptr) - 1, slot_tab.shape[0]
    slot_id = np.zeros((nout, 4), dtype=np.int32)
    slot_id[:, 2] = np.arange(nout)
    rows = [[] for _ in range(ncoef)]
    for p_ in range(nout):
        for k in range(int(ptr[p_]), int(ptr[p_ + 1])):
            rows[int(idx[k])].append((p_, float(val[k])))
    ptrT, idxT, valT = [0], [], []
    for q in range(ncoef):
        for p_, v in rows[q]:
            idxT.append(p_)
            valT.append(v)
        ptrT.append(len(idxT))
    scatter = np.full(planar_dim, -1, dtype=np.int32)
    for q in range(ncoef):
        if rows[q]:
            L, a, base, stride = (int(v) for v in slot_tab[q])
            scatter[base + a * stride] = q
    return slot_id, np.asarray(ptrT, np.int32), np.asarray(idxT, np.int32), np.asarray(valT, np.float32), scatter


def shell_block_table(row: Irreps, nao) -> np.ndarray:
    """int32[nao^2][4] = {r0, r1, c0, c1}: the (row shell, col shell) block of every matrix element (ksi block means)."""
    bounds, o = [], 0
    for _, l, _ in row:
        bounds.append((o, o + 2 * l + 1))
        o += 2 * l + 1
    assert o == nao
    owner = np.zeros(nao, dtype=np.int64)
    for b, (a0, a1) in enumerate(bounds):
        owner[a0:a1] = b
    tab = np.zeros((nao * nao, 4), dtype=np.int32)
    for r in range(nao):
        for c in range(nao):
            tab[r * nao + c] = (*bounds[owner[r]], *bounds[owner[c]])
    return tab


# ------------------------------------------------------------------------------------------------ correlation product (a21)

def corr_hidden_irreps(irreps_node, num_hidden) -> Irreps:
    """hidden irreps of CorrProductBlock (hamgnn/nn/interaction_blocks.py:196-199): num_hidden x every irrep of the node features"""
    return Irreps([(num_hidden, l, p) for _, l, p in Irreps(irreps_node)])


def sym_contraction_tables(irreps_hidden: Irreps, correlation: int = 2):
    """Sparse coupling tables of the MACE symmetric contraction with correlation <= 3 on `num_hidden x irreps`
    (hamgnn/toolbox/mace/tools/cg.py:16-131 U_matrix_real, modules/symmetric_contraction.py:101-233):
        out_k[c, w] = sum_x ( sum_kap U1_k[w, x, kap] W1_k[z, kap, c]
                              + sum_i ( sum_kap U2_k[w, x, i, kap] W2_k[z, kap, c]
                                        + sum_{j, kap} U3_k[w, x, i, j, kap] W3_k[z, kap, c] x[c, j] ) x[c, i] ) x[c, x]
    The coupling irreps are one copy of every hidden irrep, component index "ell" running over them in order.  U_nu stacks, for the
    target irrep, every coupling path in the reference's enumeration order with 'component' normalisation:
      nu = 2: (left a, right b), C = sqrt(2L+1) w3j(L, l_a, l_b);
      nu = 3: the pairs (a, b) coupled to EVERY intermediate irrep of a x b (no filter), that list sorted stably by the intermediate
              irrep (tuple order (l, p): odd before even), then the third factor c:  C = sum_m sqrt(2l_mid+1) w3j(l_mid, l_a, l_b)[m]
              sqrt(2L+1) w3j(L, l_mid, l_c)[., m, .]   (cg.py:46-87).
    Returns dict(ell_off, out_off, ptr1, ent1, ptr2, ent2, K1, K2 (per target), num_ell, nout [, ptr3, ent3, K3]) with
      ell_off[i]  planar offset of ell component i (channel 0) in the hidden layout,   out_off[o] same for output element o = (k, w)
      ent1 rows (x, kappa_global, value-bits), ent2 rows (x, i, kappa_global, value-bits), ent3 rows (x, i, j, kappa_global, value-bits);
      kappa_global indexes the concatenated weights of all targets.  ent1 / ent2 feed hg_sym_contraction, ent3 hamgnn_amd/corr3.py."""
    assert correlation in (1, 2, 3), "correlation > 3 is not built"
    lay = PlanarLayout(irreps_hidden)
    irs = [(l, p) for _, l, p in irreps_hidden]
    sl, o = [], 0
    for l, p in irs:
        sl.append((o, o + 2 * l + 1))
        o += 2 * l + 1
    num_ell = o
    ell_off = np.zeros(num_ell, np.int32)
    for j, (l, p) in enumerate(irs):
        for m in range(2 * l + 1):
            ell_off[sl[j][0] + m] = lay.off[j] + m * lay.mulp[j]
    ok = lambda l1, l2, l3: abs(l1 - l2) <= l3 <= l1 + l2
    pairs = [(a, b) for a in range(len(irs)) for b in range(len(irs))]
    # nu = 3: the (intermediate irrep, a, b) list in the order wigner_nj([.., ..]) returns it (sorted by the irrep, stably)
    left3 = sorted(((lm, irs[a][1] * irs[b][1]), a, b) for a, b in pairs for lm in range(abs(irs[a][0] - irs[b][0]), irs[a][0] + irs[b][0] + 1)) \
        if correlation >= 3 else []
    out_off, ptr1, ent1, ptr2, ent2, ptr3, ent3, K1, K2, K3 = [], [0], [], [0], [], [0], [], [], [], []
    k1g = k2g = k3g = 0
    for k, (L, pL) in enumerate(irs):
        # nu = 1: identity block of the target irrep (one path)
        paths1 = [j for j, ir in enumerate(irs) if ir == (L, pL)]
        # nu = 2: (left a, right b) with (L, pL) in a x b
        paths2 = [(a, b) for a, b in pairs if irs[a][1] * irs[b][1] == pL and ok(irs[a][0], irs[b][0], L)] if correlation >= 2 else []
        cg = {ab: math.sqrt(2 * L + 1) * so3.wigner_3j(L, irs[ab[0]][0], irs[ab[1]][0]) for ab in paths2}
        paths3 = [(mid, a, b, c) for mid, a, b in left3 for c in range(len(irs)) if mid[1] * irs[c][1] == pL and ok(mid[0], irs[c][0], L)]
        cg3 = []
        for (lm, pm), a, b, c in paths3:
            cl = math.sqrt(2 * lm + 1) * so3.wigner_3j(lm, irs[a][0], irs[b][0])              # [mid, a, b]
            cr = math.sqrt(2 * L + 1) * so3.wigner_3j(L, lm, irs[c][0])                       # [w, mid, c]
            cg3.append(np.einsum("mab,wmc->wabc", cl, cr))
        for w in range(2 * L + 1):
            out_off.append(lay.off[k] + w * lay.mulp[k])
            for kap, j in enumerate(paths1):
                ent1.append((sl[j][0] + w, k1g + kap, 1.0))
            ptr1.append(len(ent1))
            for kap, (a, b) in enumerate(paths2):
                C = cg[(a, b)][w]                                      # [2la+1, 2lb+1]
                for ma, mb in zip(*np.nonzero(np.abs(C) > 1e-14)):
                    ent2.append((sl[a][0] + ma, sl[b][0] + mb, k2g + kap, C[ma, mb]))
            ptr2.append(len(ent2))
            for kap, (mid, a, b, c) in enumerate(paths3):
                C = cg3[kap][w]
                for ma, mb, mc in zip(*np.nonzero(np.abs(C) > 1e-14)):
                    ent3.append((sl[a][0] + ma, sl[b][0] + mb, sl[c][0] + mc, k3g + kap, C[ma, mb, mc]))
            ptr3.append(len(ent3))
        K1.append(len(paths1))
        K2.append(len(paths2))
        K3.append(len(paths3))
        k1g += len(paths1)
        k2g += len(paths2)
        k3g += len(paths3)

    def pack(ents, ncols, width=4):
        arr = np.zeros((max(1, len(ents)), width), np.int32)
        for r, e in enumerate(ents):
            arr[r, :ncols] = e[:ncols]
            arr[r, width - 1] = np.float32(e[-1]).view(np.int32)
        return arr
    tab = dict(ell_off=ell_off, out_off=np.asarray(out_off, np.int32), ptr1=np.asarray(ptr1, np.int32), ent1=pack(ent1, 2),
               ptr2=np.asarray(ptr2, np.int32), ent2=pack(ent2, 3), K1=K1, K2=K2, num_ell=num_ell, nout=len(out_off))
    if correlation >= 3:
        tab.update(ptr3=np.asarray(ptr3, np.int32), ent3=pack(ent3, 4, 5), K3=K3)
    return tab

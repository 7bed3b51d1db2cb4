"""Numpy emulator of the device algorithms, consuming EXACTLY the packed buffers/tables hamgnn_amd/plan.py hands to the
HIP kernels (MFMA fragment order, segment/item records).  Dev/test tool: lets the CPU suite validate the planner and
the kernel algorithm against the oracle without a GPU.  The HIP kernels in hamgnn_amd/csrc mirror these loops."""
import math

import numpy as np

from hamgnn_amd import plan as P
from hamgnn_amd import so3

SILU_CST = 1.6791767923989418


def edge_wigner_all(n, lmax):
    """[E, sum (2l+1)^2] packed D^l(R_e), l = 0..lmax (row-major per l)."""
    offs, tot = P.wigner_offsets(lmax)
    out = np.zeros((n.shape[0], tot))
    for e in range(n.shape[0]):
        for l in range(lmax + 1):
            out[e, offs[l]:offs[l] + (2 * l + 1) ** 2] = so3.edge_wigner(l, n[e]).reshape(-1)
    return out


def rotate_rows(xp, layout, D, lmax, transpose=False):
    """x'[e][i][a][u] = sum_b D^l[a][b] x[e][i][b][u] on planar rows."""
    offs, _ = P.wigner_offsets(lmax)
    out = np.zeros_like(xp)
    for (mul, l, p), off, mp in zip(layout.irreps, layout.off, layout.mulp):
        n = 2 * l + 1
        Dl = D[:, offs[l]:offs[l] + n * n].reshape(-1, n, n)
        if transpose:
            Dl = Dl.transpose(0, 2, 1)
        blk = xp[:, off:off + n * mp].reshape(-1, n, mp)
        out[:, off:off + n * mp] = np.einsum("eab,ebu->eau", Dl, blk).reshape(-1, n * mp)
    return out


def radial_hidden(rbf, layers):
    h = rbf
    for W in layers:
        z = h @ W.astype(np.float64)
        h = z / (1 + np.exp(-z)) * SILU_CST
    return h


# csrc/tp_is.hip computes the radial scales of programs with 64 hidden units on the half-precision matrix pipe with split operands (plan/program.py:
# w3_split_fill).  S_F16 = True makes the emulator of the input-stationary schedule do the same arithmetic from the SAME table bytes
# (the split twin behind every W3 block) -- what the CPU
# stand-ins of the product's end-to-end tests run; False (default): the exact fp32-table form (the segment-stationary kernel's, and the 1e-12 planner tests').
S_F16 = False


def _split_scale(prog, Wt32, w3, rtm, hh_cols, ne, dtype):
    """S[rt][row][edge] of one item as the kernel forms it: twin block [t][rt][hi, lo][lane][4 dwords of two halves] x the hidden rows split the same way;
    SUBNORMAL halves count as zero (the half-precision MFMAs flush them), two accumulation chains, S = 2^-(sw + sh) (S0 + 2^-11 S1)"""
    n = 4 * rtm * 256
    tw = Wt32[w3 + n:w3 + 2 * n].view(np.uint32).reshape(2, rtm, 2, 64, 4)
    halves = np.stack([(tw & 0xffff).astype(np.uint16), (tw >> 16).astype(np.uint16)], -1).reshape(2, rtm, 2, 64, 8).view(np.float16).astype(dtype)   # [t][rt][term][lane][slot]
    halves[np.abs(halves) < 2.0 ** -14] = 0.0
    S0, S1 = np.zeros((rtm, 16, 16), dtype=dtype), np.zeros((rtm, 16, 16), dtype=dtype)
    for t in range(2):
        # B operand of lane (g, el), slot s: hidden unit 16 (2t + s // 4) + 4 g + s % 4 of edge el
        B = np.zeros((2, 4, 8, 16), dtype=dtype)                                          # [term][g][slot][edge]
        for g in range(4):
            for s_ in range(8):
                hi, lo = P.f16_split(hh_cols[:, 16 * (2 * t + s_ // 4) + 4 * g + s_ % 4], P.SPLIT_H_EXP)
                B[0, g, s_, :ne], B[1, g, s_, :ne] = hi.astype(dtype), lo.astype(dtype)
        B[np.abs(B) < 2.0 ** -14] = 0.0
        for rt in range(rtm):
            A = halves[t, rt].reshape(2, 4, 16, 8)                                        # [term][g][i][slot]
            S0[rt] += np.einsum("gis,gse->ie", A[0], B[0])
            S1[rt] += np.einsum("gis,gse->ie", A[0], B[1]) + np.einsum("gis,gse->ie", A[1], B[0])
    c0 = 2.0 ** -(int(getattr(prog, "w3_exp", 0)) + P.SPLIT_H_EXP)
    return c0 * S0 + (c0 * 2.0 ** -P.SPLIT_LO_EXP) * S1


def _apply_item(prog, Wt, it, srcs, h2, cols, ne, tile, lk, rto, dtype, split_scale=False):
    """one item record on one 16-edge column tile, fragment-exact (tile: [rto*16, nco, 16], updated in place / returned)."""
    E = srcs[0].shape[0]
    H = prog.hidden
    (typ, s0, s1, in_off, in_mulp, li, mm, neg, ksteps, rtm, mlp, a1, w3, cf, a2, nrows, row_off) = (int(v) for v in it[:17])
    nco = 2 * lk + 1
    if typ == P.IT_POST:                                   # tile <- Lc^T (s_e * tile), fragment-exact
        Hp = prog.hidden_pad
        W3 = Wt[w3:w3 + (Hp // 16) * rto * 256].reshape(Hp // 16, rto, 4, 16, 4)
        hh = np.zeros((E, Hp), dtype=dtype)
        hh[:, :H] = h2[0]
        S = np.zeros((rto, 16, 16), dtype=dtype)
        for G in range(Hp // 16):
            for q in range(4):
                B = np.zeros((4, 16), dtype=dtype)
                for g in range(4):
                    B[g, :ne] = hh[cols, 16 * G + 4 * g + q]
                for rt in range(rto):
                    S[rt] += W3[G, rt, :, :, q].T @ B
        A2 = Wt[a2:a2 + rto * rto * 256].reshape(rto, rto, 4, 16, 4)
        new = np.zeros_like(tile)
        for c in range(nco):
            md = tile[:, c, :].reshape(rto, 16, 16) * S
            for rtp in range(rto):
                acc = np.zeros((16, 16), dtype=dtype)
                for rt in range(rto):
                    for r in range(4):
                        acc += A2[rtp, rt, :, :, r].T @ md[rt][r::4, :]
                new[16 * rtp:16 * rtp + 16, c] = acc
        return new
    nc = 2 * mm + 1
    nsrc = 2 if s1 >= 0 else 1
    x4 = int(it[17])
    ngrp = -(-ksteps // 4)
    if typ == P.IT_LINM:                                   # one weight matrix per column (lite_mode paths of one (i, k) folded), natural K
        assert not x4 and cf == nsrc * ngrp * rtm * 256
        for c in range(nc):
            A1 = Wt[a1 + c * cf:a1 + (c + 1) * cf].reshape(nsrc, ngrp, rtm, 4, 16, 4)
            m = c - mm
            base = in_off + (li + (-m if neg else m)) * in_mulp
            for si, sidx in enumerate([s0, s1][:nsrc]):
                X = srcs[sidx]
                for G in range(ngrp):
                    for q in range(4):
                        if 4 * G + q >= ksteps:
                            continue
                        B = np.zeros((4, 16), dtype=dtype)
                        for g in range(4):
                            B[g, :ne] = X[cols, base + 4 * (4 * G + q) + g]
                        for rt in range(rtm):
                            r0 = row_off + 16 * rt
                            tile[r0:r0 + 16, lk - mm + c] += A1[si, G, rt, :, :, q].T @ B
        return tile
    A1 = Wt[a1:a1 + nsrc * ngrp * rtm * 256].reshape(nsrc, ngrp, rtm, 4, 16, 4)      # [src][G][rt][g][i][q]
    mid = np.zeros((rtm, nc, 16, 16), dtype=dtype)
    for si, sidx in enumerate([s0, s1][:nsrc]):
        X = srcs[sidx]
        for c in range(nc):
            m = c - mm
            a = li + (-m if neg else m)
            base = in_off + a * in_mulp
            for G in range(ngrp):
                for q in range(4):
                    if not x4 and 4 * G + q >= ksteps:
                        continue
                    B = np.zeros((4, 16), dtype=dtype)             # B[k = g][j = edge]
                    for g in range(4):
                        u = 16 * G + 4 * g + q if x4 else 4 * (4 * G + q) + g
                        B[g, :ne] = X[cols, base + u]
                    for rt in range(rtm):
                        mid[rt, c] += A1[si, G, rt, :, :, q].T @ B
    if typ == P.IT_TP:
        Hp = prog.hidden_pad
        W3 = Wt[w3:w3 + (Hp // 16) * rtm * 256].reshape(Hp // 16, rtm, 4, 16, 4)        # [G][rt][g][i][q]
        S = np.zeros((rtm, 16, 16), dtype=dtype)
        hh = np.zeros((E, Hp), dtype=dtype)
        hh[:, :H] = h2[mlp]
        if split_scale and Hp == 64 and any(o == w3 for o, _ in getattr(prog, "w3_regions", ())):
            S = _split_scale(prog, prog.weights, w3, rtm, hh[cols], ne, dtype)
        else:
            for G in range(Hp // 16):
                for q in range(4):
                    B = np.zeros((4, 16), dtype=dtype)
                    for g in range(4):
                        B[g, :ne] = hh[cols, 16 * G + 4 * g + q]
                    for rt in range(rtm):
                        S[rt] += W3[G, rt, :, :, q].T @ B
        CF = Wt[cf:cf + rtm * nc * 16].reshape(rtm, nc, 16)                           # [rt][c][row = 4g + r]
        mid = mid * S[:, None, :, :] * CF[:, :, :, None]
        A2 = Wt[a2:a2 + rto * rtm * 4 * 64].reshape(rto, rtm, 4, 16, 4)               # [rt'][rt][k=g][i][r]
        for rtp in range(rto):
            for c in range(nc):
                acc = np.zeros((16, 16), dtype=dtype)
                for rt in range(rtm):
                    for r in range(4):
                        if 4 * rt + r >= int(it[18]):                                # K-steps of pure padding are not issued
                            continue
                        Bm = mid[rt, c][r::4, :]                                     # rows 4k + r, k = 0..3
                        acc += A2[rtp, rt, :, :, r].T @ Bm
                tile[16 * rtp:16 * rtp + 16, lk - mm + c] += acc
    else:
        if typ == P.IT_LINC:
            mid = mid * Wt[cf:cf + nc][None, :, None, None]
        for rt in range(rtm):
            r0 = row_off + 16 * rt
            tile[r0:r0 + 16, lk - mm:lk + mm + 1] += mid[rt].transpose(1, 0, 2)
    return tile


def _write_segment(prog, seg, tile, out, cols, ne, D, woffs, dtype):
    lk, mul_k, rto, out_off, out_mulp = (int(v) for v in seg[:5])
    flags = int(seg[7])
    nco = 2 * lk + 1
    t = tile[:mul_k, :, :ne]                                                              # [w, m, e]
    if flags & P.SEG_UNROTATE:
        Dl = D[cols, woffs[lk]:woffs[lk] + nco * nco].reshape(ne, nco, nco)
        t = np.einsum("ema,wme->wae", Dl, t)                                              # out[a] = sum_m D[m][a] t[m]
    npad = (flags >> 8) & 0xff                                 # channel-padding slots the (last) chunk of a planar block zero-fills
    atomic = bool(flags & getattr(P, "SEG_ATOMIC", 2))          # one of the copies of a split segment (plan.split_heavy_segments): ADDS into zero-filled rows
    for a in range(nco):                                       # like the kernel epilogue: out[a * mulp + w], w < mul_k of THIS chunk
        o = out_off + a * out_mulp
        if atomic:
            out[np.ix_(cols, np.arange(o, o + mul_k))] += t[:, a, :].T
            continue
        out[np.ix_(cols, np.arange(o, o + mul_k))] = t[:, a, :].T
        out[np.ix_(cols, np.arange(o + mul_k, o + mul_k + npad))] = 0


def run_program(prog, srcs, h2=(None, None), D=None, lmax=None, dtype=np.float64):
    """srcs: list of planar [E, dim] arrays (slot order).  Returns planar out [E, out_layout.dim]."""
    E = srcs[0].shape[0]
    Wt = prog.weights.astype(dtype)
    out = np.zeros((E, prog.out_layout.dim), dtype=dtype)
    woffs = P.wigner_offsets(lmax)[0] if lmax is not None else None
    assert not prog.vsegs, "a program with merged items has no segment-stationary form"
    for e0 in range(0, E, 16):
        ne = min(16, E - e0)
        cols = np.arange(e0, e0 + ne)
        for seg in prog.seg_table:
            lk, mul_k, rto, out_off, out_mulp, ib, ie, flags = (int(v) for v in seg)
            tile = np.zeros((rto * 16, 2 * lk + 1, 16), dtype=dtype)
            for it in prog.item_table[ib:ie]:
                tile = _apply_item(prog, Wt, it, srcs, h2, cols, ne, tile, lk, rto, dtype)
            _write_segment(prog, seg, tile, out, cols, ne, D, woffs, dtype)
    return out


def run_program_is(prog, sched, srcs, h2=(None, None), D=None, lmax=None, dtype=np.float64):
    """the input-stationary schedule (plan.is_schedule, csrc/tp_is.hip): parts -> phases -> work groups -> items; all segment tiles of a
    part live at once in a flat LDS image [tiles | trash row | row table ...]; an item addresses its sources through the phase's staged
    blocks and the rows of its GEMM2 output through the part's ROW TABLE (item[23], item[22] row tiles) exactly as the kernel does --
    which is also how a merged item reaches the tiles of several segments.  Checks the schedule invariants on the way: every item exactly
    once, staged blocks inside the part's staging area, the LDS layout, and -- unless the part keeps a private tile copy per wave -- one
    owner work group per (phase, tile)."""
    E = srcs[0].shape[0]
    Wt = prog.weights.astype(dtype)
    out = np.zeros((E, prog.out_layout.dim), dtype=dtype)
    woffs = P.wigner_offsets(lmax)[0] if lmax is not None else None
    nprog_seg = prog.seg_table.shape[0]
    for e0 in range(0, E, 16):
        ne = min(16, E - e0)
        cols = np.arange(e0, e0 + ne)
        seen, seg_seen = set(), set()
        for sg0, nsg, ph0, nph, trash_off, stage_off, ctr_off, copy_stride, rowtab_off, rt0, rtn, _ in (tuple(int(v) for v in p[:12]) for p in sched.part_table):
            stage_floats = ctr_off - stage_off
            part_segs = set(range(sg0, sg0 + nsg))
            part_atomic = all(int(sched.seg_table[g][7]) & P.SEG_ATOMIC for g in part_segs)
            if not part_atomic:
                assert not (part_segs & seg_seen)
            seg_seen |= part_segs
            fed = set()
            tile_floats = sum(int(sched.seg_table[g][1]) * ((2 * int(sched.seg_table[g][0]) + 1) * 16 + 4) for g in part_segs)
            maxstride = max((2 * int(sched.seg_table[g][0]) + 1) * 16 + 4 for g in part_segs)
            # LDS layout of a part: [tile copies (each: tiles, trash row)] [row table] [staging area] [claim counter]
            assert trash_off == tile_floats and copy_stride in (0, tile_floats + maxstride) and (ctr_off + 4) * 4 <= P.IS_LDS_BYTES
            waves_ = P.IS_WAVES_LITE if int(sched.part_table[0][11]) else P.IS_WAVES          # wave count of the kernel instantiation (private tile copies)
            assert rowtab_off == (waves_ * copy_stride if copy_stride else tile_floats + maxstride) and stage_off == rowtab_off + rtn
            assert stage_off % 4 == 0
            rowtab = sched.rowtab[rt0:rt0 + rtn]
            lds = np.zeros(tile_floats + maxstride, dtype=dtype)               # one tile copy + its trash row
            # which tile a row-table entry points into (for the ownership check)
            tile_of = np.full(tile_floats + maxstride, -1)
            for g in part_segs:
                sgr = sched.seg_table[g]
                tile_of[int(sgr[5]):int(sgr[5]) + int(sgr[1]) * ((2 * int(sgr[0]) + 1) * 16 + 4)] = g
            for b0, b1, g0, g1 in sched.phase_table[ph0:ph0 + nph, :4]:
                staged = {}
                used = 0
                for blk in sched.block_table[b0:b1]:
                    s0, s1, in_off, in_mulp, li, nsrc, o0, o1 = (int(v) for v in blk)
                    size = -(-((2 * li + 1) * (in_mulp // 4)) // 4) * 256
                    assert o0 == used and (o1 == o0 + size if nsrc == 2 else o1 == -1)
                    used += nsrc * size
                    staged[o0] = (s0, s1, in_off, in_mulp, li)
                assert used <= stage_floats
                owner = set()
                stream_cells = set()
                if copy_stride:                                # private tile copies: the work groups are dealt to the waves (group g0 + k * waves + w), see plan._is_schedule_part
                    assert (g1 - g0) % waves_ == 0
                for gi in range(g0, g1):
                    ib, ie = (int(v) for v in sched.group_table[gi])
                    touched = set()
                    for ii in range(ib, ie):
                        it = sched.item_table[ii].copy()
                        if int(it[0]) == P.IT_STREAM:          # lite_mode stream (r4): uniform steps over the tasks (segment, row tile, column / pair) dealt to one wave
                            assert ii not in seen
                            seen.add(ii)
                            Wall = np.concatenate([prog.weights.astype(dtype), sched.extra_weights.astype(dtype)])
                            nst, w0, d0 = int(it[8]), int(it[11]), int(it[12])
                            assert nst % P.LITE_SRING == 0 and w0 % 16 == 0 and d0 % (2 * P.LITE_SRING) == 0
                            desc = Wall[d0:d0 + 2 * (nst + P.LITE_SRING)].astype(np.float32).view(np.int32)
                            assert not desc[2 * nst:].any() and not Wall[w0 + nst * 256:w0 + (nst + P.LITE_SRING) * 256].any()

                            def piece(b64):
                                """the float4 piece behind operand base b64 (64-float units into the staging area): [4 channels, 16 edges]"""
                                o0 = max(o for o in staged if o <= b64 * 64)
                                s0, s1, in_off, in_mulp, li = staged[o0]
                                size = -(-((2 * li + 1) * (in_mulp // 4)) // 4) * 256
                                src_i, rel_ = (s0, b64 * 64 - o0) if b64 * 64 - o0 < size else (s1, b64 * 64 - o0 - size)
                                assert rel_ // 64 < (2 * li + 1) * (in_mulp // 4)
                                a, s_ = divmod(rel_ // 64, in_mulp // 4)
                                B = np.zeros((4, 16), dtype=dtype)
                                for q in range(4):
                                    B[q, :ne] = srcs[src_i][cols, in_off + a * in_mulp + 4 * s_ + q]
                                return B
                            acc = accb = None
                            for t in range(nst):
                                d, e1 = int(desc[2 * t]), int(desc[2 * t + 1])
                                d, e1 = d & 0xffffffff, e1 & 0xffffffff
                                b64, nv, first, last, tc, ridx = (d >> 8) & 1023, (d & 3) + 1, (d >> 2) & 1, (d >> 3) & 1, ((d >> 18) & 31) - 16, (d >> 23) * 16
                                pair, negb, bb64, tcb = e1 >> 31, e1 & 1, (e1 >> 8) & 1023, ((e1 >> 18) & 31) - 16
                                F = Wall[w0 + t * 256:w0 + (t + 1) * 256].reshape(4, 16, 4)                          # [g][i][q]: out row i, channel 4 (4 G + q) + g
                                if first:
                                    acc = np.zeros((16, 16), dtype=dtype)
                                    accb = np.zeros((16, 16), dtype=dtype)
                                    key = (ridx, tc, tcb if pair else None)
                                if not F.any():
                                    assert not first and not last      # (padding step)
                                    continue
                                assert not F[:, :, nv:].any()                              # K-steps beyond the block's pieces are not issued
                                for q in range(nv):
                                    acc += F[:, :, q].T @ piece(b64 + q)
                                    if pair:
                                        accb += F[:, :, q].T @ (-piece(bb64 + q) if negb else piece(bb64 + q))
                                if last:
                                    for col in ([tc, tcb] if pair else [tc]):
                                        assert (ridx, col) not in stream_cells, "a (row tile, column) of a tile belongs to one task per phase"
                                        stream_cells.add((ridx, col))
                                    for i_ in range(16):
                                        base = int(rowtab[ridx + i_])
                                        lds[base + tc * 16:base + tc * 16 + 16] += acc[i_]
                                        if pair:
                                            lds[base + tcb * 16:base + tcb * 16 + 16] += accb[i_]
                                    acc = accb = None
                            assert acc is None
                            continue
                        if int(it[0]) == P.IT_POST:            # lite_mode post-op of one segment (the part's last phase, nothing staged): in place on its tile
                            assert b0 == b1 and ii not in seen
                            seen.add(ii)
                            sg = int(it[19])
                            seg = sched.seg_table[sg]
                            lk_, mul_, rto_ = int(seg[0]), int(seg[1]), int(seg[2])
                            rt = rowtab[int(it[23]):int(it[23]) + 16 * rto_]
                            tile = np.zeros((16 * rto_, 2 * lk_ + 1, 16), dtype=dtype)
                            for r in range(mul_):
                                for c in range(2 * lk_ + 1):
                                    tile[r, c] = lds[int(rt[r]) + (c - lk_) * 16:int(rt[r]) + (c - lk_) * 16 + 16]
                            new = _apply_item(prog, Wt, it[:P.ITEM_I32].copy(), srcs, h2, cols, ne, tile, lk_, rto_, dtype)
                            for r in range(mul_):
                                for c in range(2 * lk_ + 1):
                                    lds[int(rt[r]) + (c - lk_) * 16:int(rt[r]) + (c - lk_) * 16 + 16] = new[r, c]
                            touched.add(sg)
                            continue
                        s0, s1, in_off, in_mulp, li = staged[int(it[1])]
                        assert (int(it[4]), int(it[5])) == (in_mulp, li) and ((int(it[2]) >= 0) == (s1 >= 0))
                        it[1], it[2], it[3] = s0, s1, in_off
                        assert ii not in seen
                        seen.add(ii)
                        sg = int(it[19])
                        assert sg in part_segs
                        seg = sched.seg_table[sg]
                        assert tuple(int(v) for v in it[20:23])[:2] == (int(seg[0]), int(seg[1]))
                        mm, rto_i = int(it[6]), int(it[22])
                        merged = int(it[0]) == P.IT_TP and int(it[16]) > 0
                        rt = rowtab[int(it[23]):int(it[23]) + 16 * rto_i]
                        if not merged:                         # plain item: its own segment's rows, then the trash row
                            lk_, mul_, toff_ = int(seg[0]), int(seg[1]), int(seg[5])
                            strd = (2 * lk_ + 1) * 16 + 4
                            assert rto_i == int(seg[2]) and all(int(rt[r]) == toff_ + r * strd + lk_ * 16 for r in range(mul_))
                            assert all(int(rt[r]) >= trash_off for r in range(mul_, 16 * rto_i))
                        for r in range(16 * rto_i):            # every addressed row lies inside a tile (or the trash row), all 2 mm + 1 columns
                            lo, hi = int(rt[r]) - 16 * mm, int(rt[r]) + 16 * mm + 16
                            assert 0 <= lo and hi <= tile_floats + maxstride
                            assert len(set(tile_of[lo:hi].tolist())) == 1
                            touched.add(int(tile_of[lo]))
                        tmp = np.zeros((16 * rto_i, 2 * mm + 1, 16), dtype=dtype)
                        rec = it[:P.ITEM_I32].copy()
                        if merged:
                            rec[16] = 0
                        # (the kernel takes the split-half-precision form of the radial scale when the part record says the twins are there)
                        tmp = _apply_item(prog, Wt, rec, srcs, h2, cols, ne, tmp, mm, rto_i, dtype, split_scale=S_F16 and int(sched.part_table[0][12]) == 1)
                        if int(it[0]) == P.IT_TP and int(it[7]) and mm > 0:
                            # odd super-path: the kernel does not compute the centre column -- it has to vanish identically
                            assert not tmp[:, mm, :].any(), "centre column of an odd item is not structurally zero"
                        for r in range(16 * rto_i):
                            base = int(rt[r])
                            for c in range(2 * mm + 1):
                                lds[base + (c - mm) * 16:base + (c - mm) * 16 + 16] += tmp[r, c]
                    touched.discard(-1)
                    fed |= touched
                    if not copy_stride:
                        assert not (touched & owner), "a shared tile must belong to exactly one work group per phase"
                    owner |= touched
            for sg in sorted(part_segs):
                seg = sched.seg_table[sg]
                lk_, mul_, rto_, toff_ = int(seg[0]), int(seg[1]), int(seg[2]), int(seg[5])
                strd = (2 * lk_ + 1) * 16 + 4
                tile = np.zeros((rto_ * 16, 2 * lk_ + 1, 16), dtype=dtype)
                tile[:mul_] = lds[toff_:toff_ + mul_ * strd].reshape(mul_, strd)[:, :(2 * lk_ + 1) * 16].reshape(mul_, 2 * lk_ + 1, 16)
                _write_segment(prog, seg, tile, out, cols, ne, D, woffs, dtype)
        assert len(seen) == sched.item_table.shape[0] and len(seg_seen) == sched.seg_table.shape[0]
    return out


def run_linear_tables(tabs, x, res=(), dtype=np.float64):
    """csrc/linear.hip on plan.LinearTables, fragment-exact: per unit (<= 64 output channels of one irrep block) and 16 pair-rows
    (row, component), the A fragments [G][rt][lane = 16 g + i][q] hold W^T[16 rt + i][16 G + 4 g + q]; the B operand of lane (g, n) is the
    float4 x[pair-row n][16 G + 4 g .. + 3] (zero beyond the block's padded width); C[16 rt + 4 g + r][n] is stored as channel 16 rt + 4 g + r."""
    rows = x.shape[0]
    out = np.full((rows, tabs.out_dim), np.nan, dtype=dtype)
    W = tabs.weights.astype(dtype)
    for unit0, nchunks, nco, _ in (tuple(int(v) for v in g) for g in tabs.groups):
        for u in range(unit0, unit0 + nchunks):
            out_off, out_mulp, rtm, nstore, pb, pe = (int(v) for v in tabs.units[u][:6])
            assert nstore % 4 == 0 and nstore <= 16 * rtm <= 64
            acc = np.zeros((rows, nco, 16 * rtm), dtype=dtype)
            for p in range(pb, pe):
                in_off, in_mulp, ngrp, woff = (int(v) for v in tabs.paths[p])
                frag = W[woff:woff + ngrp * rtm * 256].reshape(ngrp, rtm, 4, 16, 4)          # [G][rt][g][i][q]
                Wt = frag.transpose(1, 3, 0, 2, 4).reshape(rtm * 16, ngrp * 16)                # [out channel][k = 16 G + 4 g + q]
                xin = np.zeros((rows, nco, ngrp * 16), dtype=dtype)
                xin[:, :, :in_mulp] = x[:, in_off:in_off + nco * in_mulp].reshape(rows, nco, in_mulp)
                acc += np.einsum("rak,ck->rac", xin, Wt)
            for a in range(nco):
                o = out_off + a * out_mulp
                out[:, o:o + nstore] = acc[:, a, :nstore]
    for r in res:
        out = out + r
    assert not np.isnan(out).any(), "an output column was not written"
    return out


def sym_contraction(tab, hp, z, W1, W2, C, out_dim):
    """numpy twin of hg_sym_contraction (hamgnn_amd/csrc/head.hip): hp planar hidden rows [N, Dp]; W1 [nel, K1tot, C], W2 [nel, K2tot, C]"""
    N = hp.shape[0]
    out = np.zeros((N, out_dim), dtype=np.float64)
    x = np.stack([hp[:, tab["ell_off"] + c] for c in range(C)], axis=1)          # [N, C, num_ell]
    val1 = tab["ent1"][:, 3].view(np.float32).astype(np.float64)
    val2 = tab["ent2"][:, 3].view(np.float32).astype(np.float64)
    for o in range(tab["nout"]):
        acc = np.zeros((N, C))
        for e in range(tab["ptr1"][o], tab["ptr1"][o + 1]):
            xi, kap = tab["ent1"][e, 0], tab["ent1"][e, 1]
            acc += val1[e] * W1[z, kap, :] * x[:, :, xi]
        for e in range(tab["ptr2"][o], tab["ptr2"][o + 1]):
            xi, ii, kap = tab["ent2"][e, 0], tab["ent2"][e, 1], tab["ent2"][e, 2]
            acc += val2[e] * W2[z, kap, :] * x[:, :, ii] * x[:, :, xi]
        for c in range(C):
            out[:, tab["out_off"][o] + c] = acc[:, c]
    return out


def sym_contraction3(tab, hp, z, W3, C, out):
    """numpy twin of hg_sym_contraction3 (hamgnn_amd/csrc/corr3.hip): ADDS the nu = 3 term to the planar rows `out` [N, Dp] (float64 copy returned)"""
    out = np.array(out, dtype=np.float64)
    x = np.stack([hp[:, tab["ell_off"] + c] for c in range(C)], axis=1).astype(np.float64)     # [N, C, num_ell]
    ent, ptr = tab["ent3"], tab["ptr3"]
    val = np.ascontiguousarray(ent[:, 4]).view(np.float32).astype(np.float64)
    for o in range(tab["nout"]):
        e = slice(ptr[o], ptr[o + 1])
        if e.stop == e.start:
            continue
        t = val[e][None, None, :] * np.transpose(W3[z][:, ent[e, 3], :], (0, 2, 1)) * x[:, :, ent[e, 0]] * x[:, :, ent[e, 1]] * x[:, :, ent[e, 2]]
        out[:, tab["out_off"][o] + np.arange(C)] += t.sum(-1)
    return out


def _unfrag_natural(frag, K, rows):
    """inverse of plan._frag_A(mat[K, rows], K // 4, rows // 16, x4=False) for ONE row tile: frag [G, 64, 4] -> mat [K, 16]"""
    G = frag.shape[0]
    P_ = frag.reshape(G, 4, 16, 4)                            # [G, g(lane >> 4), i(lane & 15), q]
    return P_.transpose(0, 3, 1, 2).reshape(G * 16, 16)[:K]   # k = 4 (4 G + q) + g


def run_wgrad_fused(wf, srcs, g, h2, nsplit=1, dtype=np.float64):
    """numpy twin of csrc/tp_wgrad.hip driven by the same tables (plan.WgFused): srcs = edge-frame planar source rows by slot, g = gradient rows
    of the block's output (edge frame), h2 = hidden rows of the two radial generators.  Returns (acc [nsplit, acc_floats], [gs per branch])
    in the kernel's layouts: every busy wave of every unit is followed through its own record (row tile, edge-tile lane `et`, operand
    offsets inside the staged LDS row, weight fragments, accumulator block), edge tiles are dealt to the splits as the kernel deals them."""
    from hamgnn_amd import plan as P
    E = g.shape[0]
    H = wf.hidden
    acc = np.zeros((nsplit, wf.acc_floats), dtype=dtype)
    gs = [np.zeros((E, n), dtype=dtype) for n in wf.nch]
    W = wf.weights.astype(dtype)
    T = -(-E // 16)
    for U in wf.units.astype(np.int64):
        nsrc, s0, s1, x_off, in_mulp, xp, nseg, mlp, ET, RS, PR, hp, G1 = U[:13]
        bi = int(U[15])
        slots = [s0, s1][:nsrc]
        NI = -(-T // ET)
        per = -(-NI // nsplit)
        # the staged LDS row of every edge: [x source 0 | x source 1 | gradient spans | h], exactly as wg_load / wg_store lay it out
        parts = [srcs[s][:, x_off:x_off + 4 * xp] for s in slots] + [g[:, U[16 + 2 * s]:U[16 + 2 * s] + 4 * U[17 + 2 * s]] for s in range(nseg)] + [h2[mlp][:, :H]]
        row = np.concatenate(parts, 1).astype(dtype)
        assert row.shape[1] <= RS and RS % 64 == 4 and PR * 4 == RS
        hoff = row.shape[1] - H
        hh = row[:, hoff:]
        for w in range(4):
            busy, et, nc, par, xc0, goff, g_mulp, woff, accoff, choff = U[P.WG_WREC + P.WG_WREC_I32 * w:P.WG_WREC + P.WG_WREC_I32 * (w + 1)]
            if not busy:
                continue
            G2 = -(-g_mulp // 16)
            assert P.wg_shape_ok(int(nc), int(G1), int(G2))
            comp = [(nc - 1 - c) if par else c for c in range(nc)]
            x = np.stack([np.stack([row[:, s * 4 * xp + xc0 + a * in_mulp:s * 4 * xp + xc0 + (a + 1) * in_mulp] for a in comp], 1) for s in range(nsrc)], 1)   # [E, nsrc, nc, in_mulp]
            gk = np.stack([row[:, goff + c * g_mulp:goff + (c + 1) * g_mulp] for c in range(nc)], 1)                                                              # [E, nc, g_mulp]
            o = woff
            Wm = [_unfrag_natural(W[o + s * G1 * 256:o + (s + 1) * G1 * 256].reshape(G1, 64, 4), in_mulp, 16) for s in range(nsrc)]   # [in_mulp, 16 rows]
            o += nsrc * G1 * 256
            Lm = _unfrag_natural(W[o:o + G2 * 256].reshape(G2, 64, 4), g_mulp, 16)
            o += G2 * 256
            W3m = _unfrag_natural(W[o:o + (H // 16) * 256].reshape(H // 16, 64, 4), H, 16)
            o += (H // 16) * 256
            cf = W[o:o + nc * 16].reshape(nc, 16)
            ch = wf.chtab[choff:choff + 16]
            mid = sum(np.einsum("ecu,ur->ecr", x[:, s], Wm[s]) for s in range(nsrc))          # [E, nc, 16]
            Bm = np.einsum("ecw,wr->ecr", gk, Lm)
            sv = hh @ W3m                                                                      # [E, 16]
            A_ = mid * cf[None]
            T1 = sv[:, None, :] * cf[None] * Bm
            T2 = sv[:, None, :] * A_
            for split in range(nsplit):
                its = np.arange(split * per, min(NI, (split + 1) * per))
                tiles = its * ET + et
                tiles = tiles[tiles < T]
                if tiles.size == 0:
                    continue
                rows = (tiles[:, None] * 16 + np.arange(16)[None]).reshape(-1)
                rows = rows[rows < E]
                gs[bi][np.ix_(rows, ch[ch >= 0])] = (A_[rows] * Bm[rows]).sum(1)[:, ch >= 0]   # written by the wave that owns (row tile, edge tile)
                for s in range(nsrc):
                    gW = np.einsum("ecu,ecr->ru", x[rows, s], T1[rows])                        # [16 rows, in_mulp]
                    full = np.zeros((16, G1 * 16), dtype=dtype)
                    full[:, :in_mulp] = gW
                    acc[split, accoff + s * G1 * 256:accoff + (s + 1) * G1 * 256] += full.reshape(16, G1, 16).transpose(1, 0, 2).reshape(-1)
                gL = np.einsum("ecw,ecr->rw", gk[rows], T2[rows])
                full = np.zeros((16, G2 * 16), dtype=dtype)
                full[:, :g_mulp] = gL
                acc[split, accoff + nsrc * G1 * 256:accoff + (nsrc * G1 + G2) * 256] += full.reshape(16, G2, 16).transpose(1, 0, 2).reshape(-1)
    return acc, gs


def _rp_act(x, aid):
    from hamgnn_amd import plan as P
    c = float(P.ACT_CONSTS[aid])
    if aid == P.ACT_SSP:
        return c * (np.logaddexp(0.0, x) - math.log(2.0))
    if aid == P.ACT_TANH:
        return c * np.tanh(x)
    if aid == P.ACT_SILU:
        return c * x / (1.0 + np.exp(-x))
    if aid == P.ACT_ABS:
        return c * np.abs(x)
    return x


def run_row_program(rp, x, res=(), dtype=np.float64):
    """numpy twin of csrc/rowprog.hip on plan.RowProgram tables: x [rows, din] planar rows -> [rows, dout]; res: rows added to the result.
    The units are decoded from the packed A-operand fragments (validates the packing), the LDS buffers are arrays of the planned strides."""
    from hamgnn_amd import plan as P
    rows = x.shape[0]
    buf = [np.full((rows, rp.rs[0]), np.nan, dtype=dtype), np.full((rows, rp.rs[1]), np.nan, dtype=dtype)]      # NaN: a read of something never written shows up
    buf[rp.in_buf][:, :rp.din] = x
    W = rp.weights.astype(dtype)
    for st in rp.stages.astype(np.int64):
        if st[0] == P.RP_GATE:
            b = buf[st[1]]
            act = rp.act_tab[st[3]:st[3] + st[4]]
            out = rp.out_tab[st[5]:st[5] + st[6]]
            for i, a_ in act:                                  # in place
                b[:, int(i)] = _rp_act(b[:, int(i)], int(a_))
            res_ = np.zeros((rows, int(st[6])), dtype=dtype)
            for p_, (src, gate) in enumerate(out):
                if src >= 0:
                    res_[:, p_] = b[:, src] * (b[:, gate] if gate >= 0 else 1.0)
            b[:, :int(st[6])] = res_
            continue
        src, dst = buf[st[1]], buf[st[2]]
        for w_ in range(P.RP_NW):
            for u in rp.units[st[3 + w_]:st[4 + w_]].astype(np.int64):
                in_off, in_mulp, nsteps, out_off, out_mulp, ncomp, nv4, woff, acc = u[:9]
                G = -(-nsteps // 4)
                Wm = _unfrag_natural(W[woff:woff + G * 256].reshape(G, 64, 4), 4 * nsteps, 16) if nsteps else np.zeros((0, 16))      # [K, 16 channels]
                for m in range(ncomp):
                    val = src[:, in_off + m * in_mulp:in_off + m * in_mulp + 4 * nsteps] @ Wm if nsteps else np.zeros((rows, 16))
                    sl = slice(out_off + m * out_mulp, out_off + m * out_mulp + 4 * nv4)
                    dst[:, sl] = (dst[:, sl] if acc else 0.0) + val[:, :4 * nv4]
    out = buf[rp.out_buf][:, :rp.dout].copy()
    for r in res:
        out += r
    assert np.isfinite(out).all()
    return out

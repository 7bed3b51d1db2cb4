"""Input-stationary schedule of a Program (csrc/tp_is.hip): the SAME items regrouped by input irrep block into phases, work groups and parts; lite_mode step streams."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from ._blas import single_thread_blas
from .layout import ITEM_I32, IT_LIN, IT_LINC, IT_LINM, IT_POST, IT_STREAM, IT_TP, LITE_SRING, SEG_I32, SEG_UNROTATE, ceil_div

# ---- input-stationary schedule (csrc/tp_is.hip): the SAME items, regrouped by input irrep block ------------------------------
IS_WAVES = int(os.environ.get("HG_IS_WAVES", "4"))      # waves of a workgroup (= IS_NW of csrc/tp_is.hip); all on the same 16 edges
IS_WAVES_LITE = int(os.environ.get("HG_LITE_WAVES", "8"))   # ... of the lite_mode instantiation (= IS_NW_LITE: four waves per SIMD at <= 128 VGPRs)
IS_BLOCK_I32 = 8                   # {s0, s1, in_off, in_mulp, li, nsrc, stage_off0, stage_off1}
IS_PHASE_I32 = 8                   # {block_begin, block_end, group_begin, group_end, radial generator whose hidden rows the kernel keeps resident (-1: none), 0..}
IS_LDS_BYTES = 80 * 1024           # two workgroups per CU
IS_ITEM_I32 = 24                   # item record of the IS kernel = fused-kernel record + {lk, mul_k, rto, tile_off} of its segment


@dataclass
class IsSchedule:
    seg_table: np.ndarray          # int32[nseg][8] = {lk, mul_k, rto, out_off, out_mulp, tile_off, wigner stage_off, flags | batch bit}
    block_table: np.ndarray        # int32[nblock][8]: input irrep blocks; stage offsets in floats relative to the staging area
    phase_table: np.ndarray        # int32[nphase][4]: the blocks staged together and the work groups that read them
    group_table: np.ndarray        # int32[ngroup][2] = {item_begin, item_end}: all items of one (phase, output segment); claimed
    #                                dynamically by the waves (largest first), so no two waves update one tile between barriers
    item_table: np.ndarray         # int32[nitems][24]: Program.item_table records with [1], [2] = stage offsets of source 0 / 1 (-1),
    #                                [20..23] = {lk, mul_k, rto, tile_off} of the item's segment
    part_table: np.ndarray         # int32[nparts][16] = {seg_begin, nseg, phase_begin, nphase, trash_off, stage_off, ctr_off, copy_stride,
    #                                rowtab_off (LDS float offset of the part's row table), rowtab_begin, rowtab_len, lite flag, W3 split twins (0 / 1), their exponent sw, 0, 0}
    #                                copy_stride > 0: every wave owns a private copy of the part's tiles (floats between copies)
    rowtab: np.ndarray             # int32: per part, for every (segment, row tile, row) of GEMM2's output the LDS float offset of that
    #                                row's CENTRE column (m = 0) inside its segment tile; rows beyond mul_k -> the shared trash row.
    #                                An item addresses its rows through item[23] = first table entry of its segment.
    lds_floats: int                # dynamic LDS of a workgroup (largest part)
    balance: float                 # LPT estimate: sum(cost) / (waves * sum over phases of max wave cost), worst part
    part_cost: List[int]           # estimated MFMA-slot cost of every part (critical path over its phases)
    phase_cls: List[int] = field(default_factory=list)   # per phase: the radial weight generator of its tensor-product items
    extra_weights: Optional[np.ndarray] = None          # lite_mode streams (IT_STREAM): their weight / descriptor streams, appended to Program.weights on the device
    atomic_out: bool = False                            # phase parts: the workgroups of a tile ADD their tiles into rows the host has zero-filled

    # single-part views (the common large-graph case; tests)
    @property
    def trash_off(self):
        return int(self.part_table[0][4])

    @property
    def stage_off(self):
        return int(self.part_table[0][5])

    @property
    def ctr_off(self):
        return int(self.part_table[0][6])

    @property
    def stage_floats(self):
        return self.ctr_off - self.stage_off


SEG_NEWBATCH = 1 << 16             # IS epilogue: this segment starts a new Wigner staging batch
IS_PART_I32 = 16                   # [12] = 1: the program's W3 blocks are followed by their split-half-precision twins, scaled by 2^[13] (plan/program.py:w3_split_fill); [14], [15] unused


def _item_rto(rec, segs, vsegs=()):
    """16-row tiles of GEMM2's output of an item: its segment's, or -- merged item (item[16] = virtual segment + 1) -- of all members"""
    if int(rec[0]) == IT_TP and int(rec[16]) > 0:
        return ceil_div(sum(int(segs[m][1]) for m in vsegs[int(rec[16]) - 1]), 16)
    return int(segs[int(rec[19])][2])


ITEM_OVERHEAD = int(os.environ.get("HG_ITEM_OVH", "60"))


def _item_cost(rec, segs, hp4, vsegs=()):
    if int(rec[0]) == IT_STREAM:
        return int(rec[8]) * 7 + 60
    typ, nsrc, nc, rtm = int(rec[0]), (2 if rec[2] >= 0 else 1), 2 * int(rec[6]) + 1, int(rec[9])
    c = nsrc * int(rec[8]) * rtm * nc + ITEM_OVERHEAD          # GEMM1 + a per-item latency allowance (in MFMA slots)
    if typ == IT_TP:
        c += hp4 * rtm + _item_rto(rec, segs, vsegs) * int(rec[18]) * nc
    return c


SEG_ATOMIC = 2       # (split launches) one of the copies of an output segment that share its block of the rows: the epilogue ADDS (csrc/tp_stage.h)


def lds_partition(prog: "Program") -> List[int]:
    """owner part of every output segment when the tiles of ALL segments do not fit one workgroup's LDS (the data-gradient programs:
    three feature rows of output per edge): first-fit decreasing on the tile sizes, capacity = the LDS minus the trash row, the largest
    staged input block, the row table and the claim counter.  Segments that share a work-group key (merged items) stay together."""
    nseg = prog.seg_table.shape[0]
    size = [int(s[1]) * ((2 * int(s[0]) + 1) * 16 + 4) + int(s[2]) * 16 for s in prog.seg_table]      # tile + its row-table entries
    maxstride = max((2 * int(s[0]) + 1) * 16 + 4 for s in prog.seg_table)
    need = max((2 if int(r[2]) >= 0 else 1) * ceil_div((2 * int(r[5]) + 1) * (int(r[4]) // 4), 4) * 256 for r in prog.item_table)
    cap = IS_LDS_BYTES // 4 - maxstride - need - 8 - 16 * sum(ceil_div(sum(int(prog.seg_table[m][1]) for m in v), 16) for v in prog.vsegs)
    units: Dict[int, List[int]] = {}
    for sg in range(nseg):
        units.setdefault(prog.seg_key.get(sg, sg), []).append(sg)
    bins: List[int] = []
    owner = [0] * nseg
    for key in sorted(units, key=lambda k: -sum(size[m] for m in units[k])):
        sz = sum(size[m] for m in units[key])
        if sz > cap:
            raise NotImplementedError("input-stationary schedule: one output segment's tile does not fit the LDS next to the staging area")
        for b in range(len(bins)):
            if bins[b] + sz <= cap:
                bins[b] += sz
                break
        else:
            b = len(bins)
            bins.append(sz)
        for m in units[key]:
            owner[m] = b
    return owner


@single_thread_blas
def is_schedule(prog: "Program", parts=1, separate_mlp: Optional[bool] = None) -> IsSchedule:
    """Regroup a finalized fused-kernel program for the input-stationary kernel.  Input irrep blocks (per source set) are packed
    into phases whose staged rows fit the staging area; every item reading a staged block runs in that phase.  Raises
    NotImplementedError when the tiles of all output segments + a useful staging area do not fit IS_LDS_BYTES.
    parts = "lds": the fewest parts whose tiles fit the LDS (programs with more output than one workgroup can hold).
    parts > 1: the output segments are split into `parts` sets of equal estimated cost (LPT); each set gets its own sub-schedule
    (tiles, phases, groups) and runs in its own workgroup (grid.y) -- the per-tile latency drops at the price of staging the input
    blocks once per part.  Used when a launch has fewer 16-edge tiles than the chip has workgroup slots.
    parts = ("2d", P, K) (late r5, replayed hipGraphs of the smallest crystals only: graph_capture.CapturedForward): the P segment sets of a split launch,
    each on K workgroups that take a share of the set's PHASES and ADD their tiles into zero-filled rows (segments flagged SEG_ATOMIC).  The order of those
    adds is NOT fixed -- the one schedule whose sums may differ between runs at fp32 rounding level; every other launch has one summation order (r6).
    separate_mlp: a phase only stages blocks whose tensor-product items use ONE radial weight generator (IsSchedule.phase_cls)."""
    phase_chunks = 1
    if isinstance(parts, tuple):
        assert parts[0] == "2d"
        parts, phase_chunks = int(parts[1]), max(1, int(parts[2]))
    if separate_mlp is None:
        # default: per-generator phases (the kernel re-reads its resident hidden rows once per phase and wave instead of once per generator change
        # inside a work group) when that costs less than 1 % of the estimated critical path -- programs with few phases (narrow irreps) lose
        # more balance than the re-reads cost, data-gradient and lite_mode programs have no such form
        plain = is_schedule(prog, parts if phase_chunks == 1 else ("2d", parts, phase_chunks), separate_mlp=False)
        if (parts != 1 and os.environ.get("HG_SEP_PARTS") != "1") or prog.hidden != 64:
            return plain
        try:
            sep = is_schedule(prog, parts, separate_mlp=True)
        except NotImplementedError:
            return plain
        return sep if sum(sep.part_cost) <= 1.01 * sum(plain.part_cost) else plain
    if separate_mlp and np.isin(prog.item_table[:, 0], (IT_LINC, IT_LINM, IT_POST)).any():
        raise NotImplementedError("lite_mode programs have no per-generator phases")
    hp4 = prog.hidden_pad // 4
    nseg = prog.seg_table.shape[0]
    lite_flag = int(np.isin(prog.item_table[:, 0], (IT_LINC, IT_LINM, IT_POST)).any())      # lite_mode items run in their own kernel instantiation
    s_split = int(bool(getattr(prog, "w3_regions", None)) and getattr(prog, "w3_split_ok", True))      # part record [12]: W3 blocks carry their split-half-precision twins,
    s_exp = int(getattr(prog, "w3_exp", 0))                                                              # [13]: scaled by 2^[13] (plan/program.py:w3_split_fill)
    if lite_flag and np.isin(prog.item_table[:, 0], (IT_TP, IT_LIN)).any():
        raise NotImplementedError("input-stationary schedule: a program mixes lite_mode items with tensor-product / Linear items")
    key_of = [prog.seg_key.get(sg, sg) for sg in range(nseg)]   # segments written by merged items share one work-group key
    seg_cost = np.zeros(nseg)
    for rec in prog.item_table:
        seg_cost[key_of[int(rec[19])]] += _item_cost(rec, prog.seg_table, hp4, prog.vsegs)
    owner = np.zeros(nseg, dtype=np.int64)
    nkeys = len(set(key_of))
    if parts == "lds":                                         # as few parts as the LDS allows (see lds_partition)
        owner = np.asarray(lds_partition(prog), dtype=np.int64)
        parts = int(owner.max()) + 1
    else:
        parts = max(1, min(int(parts), nkeys))
    if parts > 1 and not owner.any():
        load, held = [0.0] * parts, [0] * parts
        for sg in np.argsort(-seg_cost, kind="stable"):
            if key_of[sg] != sg:
                continue
            r = min(range(parts), key=lambda q: (load[q], held[q], q))      # (segments without items -- dead outputs -- must not pile up and leave a part empty)
            load[r] += seg_cost[sg]
            held[r] += 1
            for m in range(nseg):
                if key_of[m] == sg:
                    owner[m] = r
    # lite_mode programs with folded items: their step streams (_lite_streams) are appended to the weight blob
    runs = dict(base=int(prog.weights.size), w=[]) if (prog.item_table[:, 0] == IT_LINM).any() else None
    segs_all, btab, ptab, gtab, items_all, parttab, part_cost, rowtab_all = [], [], [], [], [], [], [], []
    phase_cls_all: List[int] = []
    lds_floats, worst_balance = 0, 1.0
    atomic_any = False
    for part in range(parts):
        members = [sg for sg in range(nseg) if owner[sg] == part]
        sub = _is_schedule_part(prog, members, hp4, seg_base=len(segs_all), block_base=len(btab), group_base=len(gtab), item_base=len(items_all),
                                split=parts > 1, separate_mlp=separate_mlp, runs=runs, waves=IS_WAVES_LITE if lite_flag else IS_WAVES)
        nph_ = len(sub["ptab"])
        K_ = min(phase_chunks, nph_) if (phase_chunks > 1 and parts > 1 and not lite_flag) else 1
        if K_ > 1:                                             # the part's phases dealt to K workgroups (LPT on the phases' critical paths), each adds its tiles
            bins_: List[List[int]] = [[] for _ in range(K_)]
            ld_ = [0] * K_
            for ph in sorted(range(nph_), key=lambda p_: -sub["phase_crit"][p_]):
                b_ = min(range(K_), key=lambda q: (ld_[q], q))
                bins_[b_].append(ph)
                ld_[b_] += sub["phase_crit"][ph] + 150
            order_ = [ph for b_ in bins_ for ph in b_]
            sub["ptab"] = [sub["ptab"][ph] for ph in order_]
            sub["phase_cls"] = [sub["phase_cls"][ph] for ph in order_]
            sub["segs"][:, 7] |= SEG_ATOMIC
            o_ = 0
            for b_ in bins_:
                parttab.append([len(segs_all), len(sub["segs"]), len(ptab) + o_, len(b_), sub["trash_off"], sub["stage_off"], sub["ctr_off"],
                                sub["copy_stride"], sub["rowtab_off"], len(rowtab_all), len(sub["rowtab"]), lite_flag, s_split, s_exp, 0, 0])
                o_ += len(b_)
            atomic_any = True
        else:
            parttab.append([len(segs_all), len(sub["segs"]), len(ptab), len(sub["ptab"]), sub["trash_off"], sub["stage_off"], sub["ctr_off"],
                            sub["copy_stride"], sub["rowtab_off"], len(rowtab_all), len(sub["rowtab"]), lite_flag, s_split, s_exp, 0, 0])
        rowtab_all += sub["rowtab"]
        phase_cls_all += sub["phase_cls"]
        segs_all += list(sub["segs"])
        btab += sub["btab"]
        ptab += sub["ptab"]
        gtab += sub["gtab"]
        items_all += list(sub["items"])
        lds_floats = max(lds_floats, sub["ctr_off"] + 4)
        worst_balance = min(worst_balance, sub["balance"])
        part_cost.append(sub["crit"])
    items = np.asarray(items_all, np.int32).reshape(-1, IS_ITEM_I32)
    return IsSchedule(np.asarray(segs_all, np.int32).reshape(-1, SEG_I32), np.asarray(btab, np.int32).reshape(-1, IS_BLOCK_I32),
                      np.asarray(ptab, np.int32).reshape(-1, IS_PHASE_I32), np.asarray(gtab, np.int32).reshape(-1, 2), items,
                      np.ascontiguousarray(np.asarray(parttab, np.int32).reshape(-1, IS_PART_I32)), np.asarray(rowtab_all, np.int32),
                      lds_floats, worst_balance, part_cost, phase_cls_all,
                      extra_weights=(np.concatenate(runs["w"]) if runs is not None and runs["w"] else None), atomic_out=atomic_any)


def _lite_column_steps(prog: "Program", items, lk: int, rtm: int, pairing: bool):
    """the column tasks of one (segment, row chunk): for every output column m (or pair +-m) the list of steps (fragment group [rtm * 256], d0, d1)
    over all folded items that feed it (descriptor words: _lite_streams).  Columns +m and -m of an (input irrep, output irrep) pair carry the SAME
    folded weight matrix up to a sign (every path of the pair has the parity of l_i + l_sh + l_k, so its aligned-frame coefficient is even or odd
    in m): a PAIRED step feeds both columns from one fragment (checked per item, not assumed), the sign rides on the second B operand; the
    centre column of an odd pair is identically zero and is not issued at all."""
    Wt = prog.weights

    def item_steps(r, m):
        so0, so1, in_mulp, li, mm, neg, ksteps, a1, colstride = int(r[1]), int(r[2]), int(r[4]), int(r[5]), int(r[6]), int(r[7]), int(r[8]), int(r[11]), int(r[13])
        if abs(m) > mm:
            return None
        c = m + mm
        nsrc, ngrp, P1 = (2 if so1 >= 0 else 1), ceil_div(ksteps, 4), in_mulp // 4
        cdir = -P1 if neg else P1
        c0p = (li - mm) * P1 + ((2 * mm) * P1 if neg else 0)
        st = []
        for si in range(nsrc):
            for G in range(ngrp):
                base = (so1 if si else so0) + (c0p + c * cdir + 4 * G) * 64
                assert base % 64 == 0 and 0 <= base // 64 < 1024
                woff = a1 + c * colstride + (si * ngrp + G) * rtm * 256
                st.append((Wt[woff:woff + rtm * 256], base // 64, min(4, ksteps - 4 * G)))
        return st

    tasks = []
    for m in range(0, lk + 1):
        cols = {sm: [item_steps(r, sm) for r in items] for sm in ((m,) if m == 0 else (m, -m))}
        paired, signs = pairing and m > 0, []
        if paired:
            for sa, sb in zip(cols[m], cols[-m]):
                if sa is None:
                    signs.append(0)
                    continue
                wa, wb = np.concatenate([x[0] for x in sa]), np.concatenate([x[0] for x in sb])
                if np.array_equal(wa, wb):
                    signs.append(1)
                elif np.array_equal(wa, -wb):
                    signs.append(-1)
                else:
                    paired = False
                    break
        if paired:
            steps = []
            for sa, sb, sg_ in zip(cols[m], cols[-m], signs):
                if sa is None or not any(np.any(x[0]) for x in sa):
                    continue
                for (w, ba, nq), (_, bb, _) in zip(sa, sb):
                    steps.append((w, (ba << 8) | (nq - 1) | ((16 + m) << 18), (bb << 8) | (1 << 31) | (1 if sg_ < 0 else 0) | ((16 - m) << 18)))
            if steps:
                tasks.append(steps)
        else:
            for mm_ in cols:
                steps = []
                for sa in cols[mm_]:
                    if sa is None or not any(np.any(x[0]) for x in sa):
                        continue
                    steps += [(w, (ba << 8) | (nq - 1) | ((16 + mm_) << 18), 0) for (w, ba, nq) in sa]
                if steps:
                    tasks.append(steps)
    return tasks


def _lite_streams(prog: "Program", recs, runs: dict, rt_base: Dict[int, int], waves: int):
    """lite_mode, input-stationary schedule, r4: ALL folded items (IT_LINM) of one phase -- `recs`, stage offsets in [1], [2] -- as `waves`
    balanced STREAMS of uniform steps, one work group each.  A TASK = (output segment, ONE 16-row tile, column m or column pair +-m): its steps
    run over every item and K group that feeds it, accumulate in registers and add into the tile once.  A step = one fragment (64 lanes x 4:
    16 output rows x up to 16 input channels = 1..4 MFMA K-steps; only the K-steps that hold channels are issued: 38 % of the steps of set-A
    feed 4 channels, a quarter of a K group) + two descriptor words
        d0 = K-steps - 1 | first step of the task << 2 | last << 3 | B operand base (in 64-float pieces) << 8 | (m + 16) << 18 | row-table index / 16 << 23 (< 255)
        d1 = 0, or for a PAIRED step: negate | B base of column -m << 8 | (-m + 16) << 18 | 1 << 31
    (fields sit where the kernel needs them with one scalar instruction each: the piece index << 8 is the operand's byte offset)
    r3 / early r4 ran one stream per (phase, segment, row chunk) with rtm row tiles per step (profiles/r03_lite.md): 147 streams per 16 edges whose first
    requests were exposed each (~37 per wave), 20 % padding steps, and per-step instruction counts that did not shrink with rtm = 1 (71 % of the
    steps).  Tasks of different segments and row tiles are independent (disjoint tile rows / columns), so the planner deals them to the waves
    by LPT on their exact step counts: 4 streams per phase, padded once each.  Returns IT_STREAM item records."""
    pairing = os.environ.get("HG_LITE_PAIR", "1") != "0"
    by_chunk: Dict[Tuple[int, int, int], list] = {}
    for r in recs:
        by_chunk.setdefault((int(r[19]), int(r[16]), int(r[9])), []).append(r)
    tasks = []                                                 # (steps [(frag 256, d0, d1)], seg)
    for (seg, row_off, rtm), items in by_chunk.items():
        lk = int(prog.seg_table[seg][0])
        assert row_off % 16 == 0
        for steps in _lite_column_steps(prog, items, lk, rtm, pairing):
            for rt in range(rtm):
                ridx = rt_base[seg] + row_off + 16 * rt
                assert ridx % 16 == 0 and ridx // 16 < 255        # (< 255: as a float32 bit pattern the word must not be a NaN -- the streams ride in the float blob)
                st = []
                for w, d0, d1 in steps:
                    f = np.asarray(w[rt * 256:(rt + 1) * 256]).reshape(4, 16, 4)       # [g][i][q]: word q of lane (g, i) = weight of channel 4 (4 G + q) + g (natural K)
                    if not np.any(f):
                        continue
                    assert not np.any(f[:, :, (d0 & 3) + 1:])                           # K-steps beyond the block's pieces carry zero weights: not issued
                    st.append((f.reshape(256), d0 | ((ridx // 16) << 23), d1))
                if st:
                    tasks.append((st, seg))
    nw = min(waves, max(1, len(tasks)))
    loads, streams = [0] * nw, [[] for _ in range(nw)]
    for st, seg in sorted(tasks, key=lambda t: -len(t[0])):
        n = loads.index(min(loads))
        loads[n] += len(st) + 2
        streams[n].append((st, seg))
    out = []
    for stream in streams:
        frags, desc = [], []
        for st, _ in stream:
            for n_, (w, d0, d1) in enumerate(st):
                frags.append(w)
                desc += [d0 | ((1 if n_ == 0 else 0) << 2) | ((1 if n_ == len(st) - 1 else 0) << 3), d1]
        nst = len(frags)
        npad = (-nst) % LITE_SRING
        # padding steps: zero weights, no task boundary; LITE_SRING more slots behind the last step (the request ring and the descriptor blocks run ahead)
        frags += [np.zeros(256)] * (npad + LITE_SRING)
        desc += [0, 0] * (npad + LITE_SRING)
        base = runs["base"] + sum(x.size for x in runs["w"])
        lead = (-base) % 16                                    # fragments and descriptor blocks on 64-byte boundaries (s_load_dwordx16)
        wblob = np.concatenate(frags).astype(np.float64)
        dblob = np.asarray(desc, dtype=np.int64).astype(np.uint32).view(np.float32).astype(np.float64)      # bit patterns (exact: float32 -> float64 -> float32)
        assert (wblob.size % 16, dblob.size % (2 * LITE_SRING)) == (0, 0) and np.array_equal(dblob.astype(np.float32).view(np.uint32), np.asarray(desc, dtype=np.int64).astype(np.uint32))
        runs["w"] += [np.zeros(lead), wblob, dblob]
        rec = np.zeros(ITEM_I32, dtype=np.int64)
        rec[0], rec[8], rec[9], rec[11], rec[12], rec[19] = IT_STREAM, nst + npad, 1, base + lead, base + lead + wblob.size, stream[0][1]
        rec[2] = -1
        out.append(rec)
    return out


def _is_schedule_part(prog: "Program", members: List[int], hp4: int, seg_base: int, block_base: int, group_base: int, item_base: int,
                      split: bool = False, separate_mlp: bool = False, runs: Optional[dict] = None, waves: int = IS_WAVES) -> dict:
    """sub-schedule of the output segments `members` (indices into prog.seg_table); all table indices are emitted as ABSOLUTE indices
    into the concatenated tables of the launch (bases given)."""
    segs = prog.seg_table[members].copy()
    local = {old: n for n, old in enumerate(members)}
    off, maxstride = 0, 0
    for s in segs:
        lk, mul_k = int(s[0]), int(s[1])
        stride = (2 * lk + 1) * 16 + 4
        s[5], s[6] = off, 0
        off += mul_k * stride
        maxstride = max(maxstride, stride)
    # split launches (several parts): every wave accumulates into its OWN copy of the part's tiles (summed before the epilogue), so the
    # items of one (phase, segment) can run on all four waves at once -- with one shared copy a part that owns one or two segments
    # would keep a single wave busy.  Taken when the four copies leave room for the largest input block.
    copy_stride = 0
    tiles_end = off + maxstride                                # one copy: the tiles, then the trash row (as wide as the widest tile)
    # lite_mode programs end with a post-op per segment (IT_POST: tile <- Lc^T (s * tile)) that runs as the part's LAST phase, on the one
    # shared copy of the tiles: no private copies then
    post_items = [rec for rec in prog.item_table if int(rec[19]) in local and int(rec[0]) == IT_POST]
    if split and not post_items:
        need = max(((2 if int(r[2]) >= 0 else 1) * ceil_div((2 * int(r[5]) + 1) * (int(r[4]) // 4), 4) * 256
                    for r in prog.item_table if int(r[19]) in local), default=0)      # (a segment nothing feeds -- structural-zero inputs -- keeps a zero tile)
        ntab = sum(int(s[2]) * 16 for s in segs) + 4 + 16 * sum(ceil_div(sum(int(prog.seg_table[m][1]) for m in v), 16) for v in prog.vsegs)
        if waves * (off + maxstride) + ntab + need + 4 <= IS_LDS_BYTES // 4:
            copy_stride = off + maxstride                      # every private copy carries its own trash row
            tiles_end = waves * copy_stride
    trash_off = off
    # row table (see IsSchedule.rowtab): offsets relative to the start of a tile copy
    lmax_part = (maxstride - 4) // 32
    rowtab: List[int] = []
    rt_base = []
    for s in segs:
        lk, mul_k, rto = int(s[0]), int(s[1]), int(s[2])
        stride = (2 * lk + 1) * 16 + 4
        rt_base.append(len(rowtab))
        rowtab += [(int(s[5]) + r * stride + lk * 16) if r < mul_k else (trash_off + lmax_part * 16) for r in range(rto * 16)]
    vt_base: Dict[int, int] = {}                               # virtual segments (merged items): the members' rows one after the other
    for rec in prog.item_table:
        v = int(rec[16]) - 1 if int(rec[0]) == IT_TP else -1
        if v < 0 or int(rec[19]) not in local or v in vt_base:
            continue
        vt_base[v] = len(rowtab)
        nrow = 0
        for m in prog.vsegs[v]:
            assert m in local, "the members of a merged item must be in one part"
            sm = segs[local[m]]
            lk, mul_k = int(sm[0]), int(sm[1])
            stride = (2 * lk + 1) * 16 + 4
            rowtab += [int(sm[5]) + r * stride + lk * 16 for r in range(mul_k)]
            nrow += mul_k
        rowtab += [trash_off + lmax_part * 16] * (ceil_div(nrow, 16) * 16 - nrow)
    rowtab += [0] * ((-len(rowtab)) % 4)
    rowtab_off = tiles_end
    stage_off = rowtab_off + len(rowtab)
    stage_floats = IS_LDS_BYTES // 4 - stage_off - 4
    # ---- input blocks read by this part's items
    blocks: Dict[Tuple[int, int, int], dict] = {}
    for rec in prog.item_table:
        if int(rec[19]) not in local or int(rec[0]) == IT_POST:
            continue
        key = (int(rec[1]), int(rec[2]), int(rec[3]))
        if int(rec[5]) > 6:
            raise NotImplementedError("input irreps with l > 6 have no staging instantiation")
        b = blocks.setdefault(key, dict(key=key, in_mulp=int(rec[4]), li=int(rec[5]), items=[]))
        assert b["in_mulp"] == int(rec[4]) and b["li"] == int(rec[5])
        b["items"].append(rec)
    for b in blocks.values():
        b["nsrc"] = 2 if b["key"][1] >= 0 else 1
        cls = {int(r[10]) for r in b["items"] if int(r[0]) == IT_TP}
        if separate_mlp and len(cls) > 1:                      # (data-gradient programs: one staged gradient block feeds both branches)
            raise NotImplementedError("static-stream schedule: an input block whose items use both radial weight generators")
        b["cls"] = cls.pop() if len(cls) == 1 else None        # None: plain Linear items only (no radial scale) / not separated
        b["src_floats"] = ceil_div((2 * b["li"] + 1) * (b["in_mulp"] // 4), 4) * 256
        b["floats"] = b["nsrc"] * b["src_floats"]
        if b["floats"] > stage_floats:
            raise NotImplementedError(f"input-stationary schedule: LDS staging area of {stage_floats * 4} B is smaller than an input block")
    # the staging area only needs to hold the largest phase: parts with small tiles keep the LDS small as well
    # ---- phases: first-fit decreasing packing of the blocks into the staging area
    phases: List[List[dict]] = []
    def _cls(ph):
        return next((x["cls"] for x in ph if x["cls"] is not None), None)

    for b in sorted(blocks.values(), key=lambda b: -b["floats"]):
        for ph in phases:
            if sum(x["floats"] for x in ph) + b["floats"] <= stage_floats and (
                    not separate_mlp or b["cls"] is None or _cls(ph) is None or _cls(ph) == b["cls"]):
                ph.append(b)
                break
        else:
            phases.append([b])
    btab, ptab, gtab, items, tot, crit = [], [], [], [], 0, 0
    phase_cls: List[int] = []
    phase_crit: List[int] = []                                 # per phase: the dearest wave's load (phase parts, see is_schedule)
    phase_touch: List[set] = []                                # per phase: the output segments (program indices) its items write
    for ph in phases:
        ph.sort(key=lambda b: -b["key"][0])                    # edge-row blocks (plain LDS-DMA) first: their latency runs under the
        b0, g0, o = len(btab), len(gtab), 0                    # rotation work of the node-row blocks
        by_seg: Dict[int, List[np.ndarray]] = {}
        for b in ph:
            s0, s1, in_off = b["key"]
            o0, o1 = o, (o + b["src_floats"] if b["nsrc"] == 2 else -1)
            o += b["floats"]
            btab.append([s0, s1, in_off, b["in_mulp"], b["li"], b["nsrc"], o0, o1])
            for rec in b["items"]:
                r = rec.copy()
                r[1], r[2], r[3] = o0, o1, 0
                by_seg.setdefault(prog.seg_key.get(int(rec[19]), int(rec[19])), []).append(r)
        if copy_stride:                                        # private tile copies: every item is its own work group
            units = [[r] for recs in by_seg.values() for r in recs]
        else:                                                  # a work group's items by radial generator: the kernel keeps the hidden rows of
            units = [sorted(recs, key=lambda r: int(r[10]) if int(r[0]) == IT_TP else -1) for recs in by_seg.values()]      # ONE generator in registers
        if runs is not None and all(int(r[0]) == IT_LINM for recs in units for r in recs):
            # the phase's folded items as `waves` balanced streams of uniform steps, one work group each (disjoint (row tile, column) cells of the tiles)
            units = [[st] for st in _lite_streams(prog, [r for recs in units for r in recs], runs, {sg: rt_base[n] for sg, n in local.items()}, waves)]
        groups = sorted(((sum(_item_cost(r, prog.seg_table, hp4, prog.vsegs) for r in recs), n) for n, recs in enumerate(units)), reverse=True)
        loads = [0] * waves
        dealt: List[List[int]] = [[] for _ in range(waves)]    # LPT: the work groups of every wave, dearest first
        for c, n in groups:                                    # claim order = LPT order
            w_ = loads.index(min(loads))
            loads[w_] += c
            dealt[w_].append(n)
            if not copy_stride:
                gtab.append([item_base + len(items), item_base + len(items) + len(units[n])])
                items += units[n]
        if copy_stride:
            # private tile copies (r6): WHICH wave adds an item into WHICH copy is fixed here, not by the claim order of a run -- the copies are folded in a
            # fixed order, so with static dealing the whole launch has one summation order (the dynamic claims of r2-r5 made two forwards of a small crystal
            # differ at fp32 rounding level: VERDICT r5).  Group g0 + k * waves + w is the k-th work group of wave w; short streams end with empty groups.
            for k in range(max(len(d) for d in dealt)):
                for w_ in range(waves):
                    if k < len(dealt[w_]):
                        n = dealt[w_][k]
                        gtab.append([item_base + len(items), item_base + len(items) + len(units[n])])
                        items += units[n]
                    else:
                        gtab.append([item_base + len(items), item_base + len(items)])
        tot += sum(loads)
        crit += max(loads)
        phase_crit.append(max(loads))
        phase_touch.append({int(m_) for recs in units for r in recs for m_ in ([int(r[19])] if not (int(r[0]) == IT_TP and int(r[16]) > 0) else prog.vsegs[int(r[16]) - 1])})
        # the generator whose hidden rows stay in registers during the phase: the one that carries most of its tensor-product work
        w = [sum(_item_cost(r, prog.seg_table, hp4, prog.vsegs) for recs in units for r in recs if int(r[0]) == IT_TP and int(r[10]) == c) for c in (0, 1)]
        res_cls = -1 if not any(w) else int(w[1] > w[0])
        ptab.append([block_base + b0, block_base + len(btab), group_base + g0, group_base + len(gtab), res_cls, 0, 0, 0])
        phase_cls.append(_cls(ph) or 0)
    if post_items:                                             # the last phase: nothing staged, one work group per segment's post-op, dearest first
        g0 = len(gtab)
        pc = lambda r: int(prog.seg_table[int(r[19])][2]) * (hp4 + int(prog.seg_table[int(r[19])][2]) * 4 * (2 * int(prog.seg_table[int(r[19])][0]) + 1)) + 60
        loads = [0] * waves
        for rec in sorted(post_items, key=lambda r: -pc(r)):
            r = rec.copy()
            r[1], r[2], r[3] = 0, -1, 0
            loads[loads.index(min(loads))] += pc(rec)
            gtab.append([item_base + len(items), item_base + len(items) + 1])
            items.append(r)
        tot += sum(loads)
        crit += max(loads)
        ptab.append([block_base + len(btab), block_base + len(btab), group_base + g0, group_base + len(gtab), -1, 0, 0, 0])
        phase_cls.append(0)
    # ---- epilogue: Wigner blocks of the un-rotated segments staged in as few batches as fit the staging area (one block per l)
    need = {}
    for sg in segs:
        if int(sg[7]) & SEG_UNROTATE:
            need[int(sg[0])] = ceil_div((2 * int(sg[0]) + 1) ** 2, 4) * 64
    batches: List[List[int]] = []
    wig_cap = stage_floats
    for l in sorted(need, key=lambda l: -need[l]):
        if need[l] > wig_cap:
            raise NotImplementedError("input-stationary schedule: staging area smaller than a Wigner block")
        for bt in batches:
            if sum(need[x] for x in bt) + need[l] <= wig_cap:
                bt.append(l)
                break
        else:
            batches.append([l])
    woff, batch_of = {}, {}
    for bi, bt in enumerate(batches):
        o = 0
        for l in bt:
            woff[l], batch_of[l] = o, bi
            o += need[l]
    order = sorted(range(len(segs)), key=lambda i: (batch_of.get(int(segs[i][0]), -1) if int(segs[i][7]) & SEG_UNROTATE else -1))
    remap = {members[old]: seg_base + new for new, old in enumerate(order)}
    segs2 = segs[order].copy()
    prev = None
    for sg in segs2:
        if int(sg[7]) & SEG_UNROTATE:
            l = int(sg[0])
            sg[6] = woff[l]
            if batch_of[l] != prev:
                sg[7] |= SEG_NEWBATCH
                prev = batch_of[l]
    items = np.asarray(items, np.int32).reshape(-1, ITEM_I32)
    wide = np.zeros((items.shape[0], IS_ITEM_I32), np.int32)   # + the segment fields an item needs (csrc/tp_is.hip)
    wide[:, :ITEM_I32] = items
    for n in range(items.shape[0]):
        g_abs = remap[int(items[n, 19])]
        sg = segs2[g_abs - seg_base]
        wide[n, 19] = g_abs
        wide[n, 20], wide[n, 21], wide[n, 22], wide[n, 23] = sg[0], sg[1], sg[2], rt_base[order[g_abs - seg_base]]
        v = int(items[n, 16]) - 1 if int(items[n, 0]) == IT_TP else -1
        if v >= 0:                                             # merged item: rows of all members, through the virtual segment's table range
            wide[n, 22], wide[n, 23] = _item_rto(items[n], prog.seg_table, prog.vsegs), vt_base[v]
    ctr_off = stage_off + stage_floats
    return dict(segs=segs2.astype(np.int32), btab=btab, ptab=ptab, gtab=gtab, items=wide, trash_off=trash_off, stage_off=stage_off, remap=remap,
                phase_crit=phase_crit, phase_touch=phase_touch, phase_cls=phase_cls, rowtab=rowtab, rowtab_off=rowtab_off, ctr_off=ctr_off, copy_stride=copy_stride, balance=tot / (waves * crit) if crit else 1.0, crit=crit)

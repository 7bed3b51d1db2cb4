// tp_st.hip -- "static stream" schedule of the fused equivariant edge kernel (gfx950, CDNA4).  Hand-written HIP.
//
// Same algebra, same LDS picture and the same staging / epilogue as csrc/tp_is.hip (a workgroup = 4 waves = ONE tile of 16 edges, the
// tiles of all output segments resident in LDS, input irrep blocks staged phase by phase), but the work inside a phase is no longer
// claimed dynamically item by item with every item fetching its own weight fragments behind four dependent L2 round trips
// (profiles/r02_tp_is_experiments.md section 1: -15 % with the weight loads ablated, 65 items per wave and tile).  Instead
//   * the planner (hamgnn_amd/plan.py:st_schedule) assigns the work groups of every phase to the four waves STATICALLY (the LPT order the
//     dynamic claim converged to anyway), so the sequence of weight fragments a wave consumes over the whole 16-edge pass is known at
//     plan time and is laid out as three contiguous per-wave streams:  A (GEMM1 / GEMM2 fragments, 1 KiB each, consumption order),
//     R (last radial layer, 4 fragments per 16-row tile), C (aligned-frame CG coefficients);
//   * a wave walks its streams through a register FIFO that is always ST_D fragments ahead -- across row tiles, items and phases -- and
//     the radial fragments of the NEXT row tile are requested while the current one computes: no weight latency is exposed at an item
//     boundary, whatever the item's shape;
//   * the unit of work is ONE 16-row tile of a super-path (radial scale -> GEMM1 -> scale -> GEMM2), so the carried state between
//     template instantiations has a fixed size; GEMM2 accumulates an item's output rows in registers ACROSS its row tiles (initialised
//     from the LDS tile when the item starts, written back once when it ends) instead of a read-modify-write per row-tile step.
// Row tiles, fragments and coefficients are those of the planner's items (same MFMA count as tp_is).
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef HG_STAGE_FUSE_LMAX_ST
#define HG_STAGE_FUSE_LMAX_ST 3   // both node sources rotated in one pass up to this l (the carried stream state lives through the staging)
#endif
#define HG_STAGE_FUSE_LMAX HG_STAGE_FUSE_LMAX_ST
#include "tp_stage.h"

#define ST_OP_I32 16             // plan.py:ST_OP_I32
#ifndef ST_RADAHEAD
#define ST_RADAHEAD 0             // 1: radial scale computed one row tile ahead, beside the previous tile's scale step (measured slower: 8.81 vs 8.01 ms, profiles/r03_tp_st_experiments.md)
#endif

// phase profiler (HG_PROF builds only, tests/bench_tp.py): per-wave shader-clock time between probes, summed over waves
#ifdef HG_PROF
__device__ unsigned long long hg_prof_st_acc[16];
#define ST_T(k)                                                      \
    do {                                                             \
        const unsigned long long t_ = __builtin_readcyclecounter();  \
        s.t[k] += t_ - s.last;                                       \
        s.last = t_;                                                 \
    } while (0)
#else
#define ST_T(k)
#endif

// op record (int32[16]): {code, so0, so1, fb0, cdir64, ngrp, ksteps, nsrc, rtm, rto, nk2, flags (1 mlp, 2 linear, 4 permuted-K),
//                         rowtab index, row0, item index, 0}
// State a wave carries from row tile to row tile, across items and phases: where it stands in its three streams, the radial fragments
// of the row tile it runs next (requested one row tile ahead) and the hidden rows of the phase's weight generator.
struct StState {
    const f32x4* pa;             // A stream: next fragment to REQUEST (per-lane pointer)
    f32x4 wr[4];                 // radial fragments of the next row tile to run
    const f32x4* pr;             // R stream: next fragment group to request
    const f32x4* pc;             // C stream (per-lane pointer, lane group g folded in)
    f32x4 hb[4];                 // radial hidden rows of the 16 edges (the weight generator of the CURRENT phase) as MFMA B operands
    f32x4 S;                     // radial scale of the row tile about to run (computed one row tile ahead)
#ifdef HG_PROF
    unsigned long long t[12], last;
#endif
};

// fragments per GEMM1 step: a step's MFMAs (4 NC per fragment, 32 cycles each) have to cover the L2 latency of the NEXT step's
// fragments, which are requested when the step starts: 4 x 128 cycles at one column, 2 x 384 at three, >= 640 beyond
template <int MM> struct StStep { static constexpr int FS = MM == 0 ? 4 : (MM == 1 ? 2 : 1); };

// ---- B operands of GEMM1 (staged block columns), software-pipelined: the operands of fragment t + 1 are requested from LDS before the
// MFMAs of fragment t are issued.  X4 (permuted K: the block's channel count is a multiple of 16 and NC <= 3): the operand of lane
// (g, el) for the four K-steps of a fragment is ONE float4 of the staged image; otherwise natural K: one dword per K-step.
// t = flat (source, K group) index of the fragment.
template <int MM, bool ODD, bool X4> struct StB {
    static constexpr int NC = ODD ? 2 * MM : 2 * MM + 1;
#ifndef ST_BPIPE
#define ST_BPIPE 0               // 1: operands of fragment t + 1 requested from LDS before the MFMAs of fragment t (measured neutral: 8.04 vs 8.01 ms)
#endif
    static constexpr bool PIPE = ST_BPIPE && (X4 || MM <= 2);  // natural K beyond l = 2: no second operand set in the register budget
    f32x4 v[X4 ? NC : 1];
    float d[X4 ? 1 : (MM <= 2 ? 4 : 1)][X4 ? 1 : NC];
};
#define ST_COL(c) ((ODD && (c) >= MM) ? (c) + 1 : (c))

template <int MM, bool ODD, bool X4>
__device__ __forceinline__ void st_bread(StB<MM, ODD, X4>& b, const float* __restrict__ stage, int so0, int so1, int fb0, int cdir64, int ngrp,
                                         int t, int g, int el) {
    constexpr int NC = StB<MM, ODD, X4>::NC;
    const int si = t >= ngrp ? 1 : 0, G = t - si * ngrp;
    const float* __restrict__ sbase = stage + (si ? so1 : so0) + fb0 + G * 256;
    if constexpr (X4) {                                        // fragment (c, G) = piece c0p + 4 G + g of row el
        const float* __restrict__ fb = sbase + g * 64 + el * 4;
#pragma unroll
        for (int c = 0; c < NC; ++c) b.v[c] = *reinterpret_cast<const f32x4*>(fb + ST_COL(c) * cdir64);
    } else {                                                   // element (c, 4 sl + g) = piece c0p + sl, component g
        const float* __restrict__ fb = sbase + el * 4 + g;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) b.d[q][c] = fb[ST_COL(c) * cdir64 + q * 64];      // (K-steps beyond the block's channels: read, not used)
    }
}

// MFMAs of one fragment on operands already in registers (PIPE) or read here (natural K, l >= 3)
template <int MM, bool ODD, int NACC, bool X4>
__device__ __forceinline__ void st_gemm1_frag(const f32x4 a, const StB<MM, ODD, X4>& b, f32x4 (&mid)[NACC][ODD ? 2 * MM : 2 * MM + 1],
                                              const float* __restrict__ stage, int so0, int so1, int fb0, int cdir64, int ngrp, int ksteps,
                                              int t, int g, int el) {
    constexpr int NC = StB<MM, ODD, X4>::NC;
    const int si = t >= ngrp ? 1 : 0, G = t - si * ngrp;
    if constexpr (X4) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) mid[q % NACC][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b.v[c][q], mid[q % NACC][c], 0, 0, 0);
    } else if constexpr (MM <= 2) {
        const int nq = ksteps - 4 * G;                         // K-steps of this fragment: 4, fewer in a source's last group
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nq) {
#pragma unroll
                for (int c = 0; c < NC; ++c) mid[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b.d[q][c], mid[0][c], 0, 0, 0);
            }
    } else {
        const float* __restrict__ fb = stage + (si ? so1 : so0) + fb0 + G * 256 + el * 4 + g;
        const int nq = ksteps - 4 * G;
#define ST_KSTEP(q)                                                                                                           \
        {                                                                                                                     \
            float bb[NC];                                                                                                     \
            _Pragma("unroll") for (int c = 0; c < NC; ++c) bb[c] = fb[ST_COL(c) * cdir64 + (q) * 64];                         \
            _Pragma("unroll") for (int c = 0; c < NC; ++c) mid[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], bb[c], mid[0][c], 0, 0, 0); \
        }
        if constexpr (MM >= 4) {                               // wide items read blocks of <= 12 channels (<= 3 K-steps): compact counted loop
#pragma unroll 1
            for (int q = 0; q < nq && q < 4; ++q) {
                float bb[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) bb[c] = fb[ST_COL(c) * cdir64 + q * 64];
                const float aq = q == 0 ? a[0] : (q == 1 ? a[1] : (q == 2 ? a[2] : a[3]));
#pragma unroll
                for (int c = 0; c < NC; ++c) mid[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bb[c], mid[0][c], 0, 0, 0);
            }
        } else if (nq >= 4) {                                  // l = 3: operands of two K-steps requested together
#pragma unroll
            for (int q0 = 0; q0 < 4; q0 += 2) {
                float bb[2][NC];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < NC; ++c) bb[q][c] = fb[ST_COL(c) * cdir64 + (q0 + q) * 64];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int c = 0; c < NC; ++c) mid[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q0 + q], bb[q][c], mid[0][c], 0, 0, 0);
            }
        } else if (nq == 3) {
            ST_KSTEP(0) ST_KSTEP(1) ST_KSTEP(2)
        } else if (nq == 2) {
            ST_KSTEP(0) ST_KSTEP(1)
        } else {
            ST_KSTEP(0)
        }
#undef ST_KSTEP
    }
}

// radial scale s_e = W3^T h2 of the 16 rows of the row tile whose fragments are in s.wr (K = 64 hidden units, two partial accumulators);
// then the fragments of the row tile after that are requested
__device__ __forceinline__ f32x4 st_radial(StState& s) {
    f32x4 S0 = (f32x4){0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
    for (int G = 0; G < 4; ++G)
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
            S0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s.wr[G][q], s.hb[G][q], S0, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s.wr[G][q + 1], s.hb[G][q + 1], S1, 0, 0, 0);
        }
#pragma unroll
    for (int G = 0; G < 4; ++G) s.wr[G] = s.pr[G * 64];
    s.pr += 256;
    return S0 + S1;
}

// plain o3.Linear item (PairInteractionBlock skip): rows are output channels, GEMM1 only, added straight into the tile
template <int MM, bool X4>
__device__ __forceinline__ void item_lin(const IsArgs& A, const int* __restrict__ op, float* __restrict__ lds, int lane, StState& s) {
    asm volatile("" : "+v"(lane));                             // per-item address arithmetic stays inside the item (see item_st)
    constexpr int NC = 2 * MM + 1;
    constexpr int FS = StStep<MM>::FS;
    constexpr bool ODD = false;
    const int so0 = op[1], fb0 = op[3], cdir64 = op[4], ngrp = op[5], ksteps = op[6], rtm = op[8], row0 = op[13];
    const int g = lane >> 4, el = lane & 15;
    const int* __restrict__ rtab = reinterpret_cast<const int*>(lds + A.rowtab_off) + op[12];
    float* __restrict__ tbase = lds + (el - MM * 16);
    const float* __restrict__ stage = lds + A.stage_off;
    f32x4 an[FS];
#pragma unroll
    for (int j = 0; j < FS; ++j) an[j] = s.pa[j * 64];         // (reads ahead of the item when it has fewer fragments: the stream is padded)
    StB<MM, false, X4> bn;
    if (StB<MM, false, X4>::PIPE) st_bread<MM, false, X4>(bn, stage, so0, so0, fb0, cdir64, ngrp, 0, g, el);
#pragma unroll 1
    for (int rt = 0; rt < rtm; ++rt) {
        f32x4 mid[1][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) mid[0][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int t0 = 0; t0 < ngrp; t0 += FS) {
            f32x4 a[FS];
#pragma unroll
            for (int j = 0; j < FS; ++j) a[j] = an[j];
            const int used = ngrp - t0 < FS ? ngrp - t0 : FS;
            s.pa += used * 64;
#pragma unroll
            for (int j = 0; j < FS; ++j) an[j] = s.pa[j * 64];
#pragma unroll
            for (int j = 0; j < FS; ++j)
                if (j < used) {
                    StB<MM, false, X4> bc = bn;
                    if (StB<MM, false, X4>::PIPE) st_bread<MM, false, X4>(bn, stage, so0, so0, fb0, cdir64, ngrp, t0 + j + 1 < ngrp ? t0 + j + 1 : 0, g, el);
                    else if (X4 || MM <= 2) st_bread<MM, false, X4>(bc, stage, so0, so0, fb0, cdir64, ngrp, t0 + j, g, el);
                    st_gemm1_frag<MM, false, 1, X4>(a[j], bc, mid, stage, so0, so0, fb0, cdir64, ngrp, ksteps, t0 + j, g, el);
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* __restrict__ t0p = tbase + rtab[row0 + 16 * rt + 4 * g + r];
#pragma unroll
            for (int c = 0; c < NC; ++c) t0p[c * 16] += mid[0][c][r];
        }
    }
    ST_T(10);                                                  // Linear items
}

// One tensor-product item = the row tiles of one super-path chunk, software-pipelined:
//   row tile rt:  GEMM1(rt)  ->  [radial(rt + 1) MFMAs | scale(rt) on the VALU]  ->  GEMM2(rt)
// s.S holds the radial scale of the row tile about to run (computed one row tile ahead, across item boundaries inside a work group);
// `last`: the group's last item (no row tile follows its last one).
template <int MM, int RTO, bool ODD, bool X4>
__device__ __forceinline__ void item_st(const IsArgs& A, const int* __restrict__ op, float* __restrict__ lds, int lane, StState& s, bool last) {
    // opaque copy of the lane id: without it the compiler hoists the lane-derived address terms of ALL ~30 instantiations out of the
    // op loop and keeps them live through the whole kernel (78 spilled VGPRs, reloaded from scratch inside the items)
    asm volatile("" : "+v"(lane));
    constexpr int NCR = 2 * MM + 1;                            // real columns
    constexpr int NC = ODD ? 2 * MM : NCR;                     // column slots (odd super-paths: the centre column vanishes)
    constexpr int NACC = (NC == 1 && X4) ? 2 : 1;              // one column: two partial accumulators break the dependent MFMA chain
    constexpr int FS = StStep<MM>::FS;
    constexpr bool PIPE = StB<MM, ODD, X4>::PIPE;
    const int so0 = op[1], so1 = op[2], fb0 = op[3], cdir64 = op[4], ngrp = op[5], ksteps = op[6], nsrc = op[7], rtm = op[8];
    const int rto = op[9], nk2 = op[10];
    const int g = lane >> 4, el = lane & 15;
    const int* __restrict__ rtab = reinterpret_cast<const int*>(lds + A.rowtab_off) + op[12];
    float* __restrict__ tbase = lds + (el - MM * 16);
    const float* __restrict__ stage = lds + A.stage_off;
    const int nA = nsrc * ngrp;
    ST_T(0);                                                   // dispatch (op record, switch)

    // the first GEMM1 step's fragments and the first fragment's B operands: requested before anything else
    f32x4 an[FS];
#pragma unroll
    for (int j = 0; j < FS; ++j) an[j] = s.pa[j * 64];         // (reads ahead when nA < FS: the stream is padded)
    StB<MM, ODD, X4> bn;
    if (PIPE) st_bread<MM, ODD, X4>(bn, stage, so0, so1, fb0, cdir64, ngrp, 0, g, el);
    // ---------------------------------------------------------------- GEMM2 accumulators = the item's output rows, from the LDS tile
    f32x4 acc[RTO][NC];
#pragma unroll
    for (int rtp = 0; rtp < RTO; ++rtp)
        if (rtp < rto) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* __restrict__ t = tbase + rtab[16 * rtp + 4 * g + r];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[rtp][c][r] = t[ST_COL(c) * 16];
            }
        }
    ST_T(1);                                                   // accumulator init

#pragma unroll 1
    for (int rt = 0; rt < rtm; ++rt) {
        // coefficients of this row tile: requested first, used after GEMM1 (wide items, MM >= 3: requested after GEMM1 -- their three
        // NC-wide register arrays do not fit next to the carried stream state)
        constexpr bool LATECF = MM >= 3;
        f32x4 cfv[NC];
        if (!LATECF) {
#pragma unroll
            for (int c = 0; c < NC; ++c) cfv[c] = s.pc[ST_COL(c) * 4];
        }
        // ------------------------------------------------------------ GEMM1: mid = A1 fragments x staged block, FS fragments per step;
        // the next step's fragments -- or, from the last step, GEMM2's -- are requested when a step starts
        f32x4 mid[NACC][NC];
#pragma unroll
        for (int h = 0; h < NACC; ++h)
#pragma unroll
            for (int c = 0; c < NC; ++c) mid[h][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 a2n[RTO];
#pragma unroll 1
        for (int t0 = 0; t0 < nA; t0 += FS) {
            f32x4 a[FS];
#pragma unroll
            for (int j = 0; j < FS; ++j) a[j] = an[j];
            const int used = nA - t0 < FS ? nA - t0 : FS;
            s.pa += used * 64;
            if (t0 + FS < nA) {
#pragma unroll
                for (int j = 0; j < FS; ++j) an[j] = s.pa[j * 64];
            } else {
#pragma unroll
                for (int rtp = 0; rtp < RTO; ++rtp) a2n[rtp] = s.pa[rtp * 64];
            }
#pragma unroll
            for (int j = 0; j < FS; ++j)
                if (j < used) {
                    StB<MM, ODD, X4> bc = bn;                  // (the B operands do not depend on the row tile: after the last fragment, the first again)
                    if (PIPE) st_bread<MM, ODD, X4>(bn, stage, so0, so1, fb0, cdir64, ngrp, t0 + j + 1 < nA ? t0 + j + 1 : 0, g, el);
                    else if (X4 || MM <= 2) st_bread<MM, ODD, X4>(bc, stage, so0, so1, fb0, cdir64, ngrp, t0 + j, g, el);
                    st_gemm1_frag<MM, ODD, NACC, X4>(a[j], bc, mid, stage, so0, so1, fb0, cdir64, ngrp, ksteps, t0 + j, g, el);
                }
        }
        ST_T(2);                                               // GEMM1
        // ------------------------------------------------------------ radial scale of the NEXT row tile (MFMA pipe) beside the scale of
        // this one (VALU): mid *= s_e * coefficient
#ifndef ST_RADAHEAD
#define ST_RADAHEAD 1
#endif
#if ST_RADAHEAD
        const f32x4 S = s.S;
        if (!(last && rt + 1 == rtm)) s.S = st_radial(s);
#else
        const f32x4 S = st_radial(s);
#endif
        {
            // wide items: the coefficients arrive in batches of CH columns (register budget; l >= 5: 18 of 260 items of set-A)
            constexpr int CH = MM >= 6 ? 4 : (MM >= 5 ? (NC + 1) / 2 : NC);
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += CH) {
                if (LATECF) {
                    if (c0) asm volatile("" ::: "memory");
#pragma unroll
                    for (int c = c0; c < c0 + CH && c < NC; ++c) cfv[c - c0] = s.pc[ST_COL(c) * 4];
                }
#pragma unroll
                for (int c = c0; c < c0 + CH && c < NC; ++c) {
                    if (NACC == 2) mid[0][c] += mid[NACC - 1][c];
                    mid[0][c] = mid[0][c] * (S * cfv[LATECF ? c - c0 : c]);
                }
            }
            s.pc += NCR * 4;
        }
        ST_T(3);                                               // radial (next row tile) + scale
        // ------------------------------------------------------------ GEMM2: acc[w'', m] += L' fragments x mid (all rto fragments of the row
        // tile were requested by the last GEMM1 step); K-steps that hold only padding rows are not issued.  The first GEMM1 step of the
        // next row tile is requested now
        f32x4 a2[RTO];
#pragma unroll
        for (int rtp = 0; rtp < RTO; ++rtp) a2[rtp] = a2n[rtp];
        s.pa += rto * 64;
        if (rt + 1 < rtm) {
#pragma unroll
            for (int j = 0; j < FS; ++j) an[j] = s.pa[j * 64];
        }
        const int kv = nk2 - 4 * rt;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < kv) {
#pragma unroll
                for (int rtp = 0; rtp < RTO; ++rtp)
                    if (rtp < rto) {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            acc[rtp][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[rtp][r], mid[0][c][r], acc[rtp][c], 0, 0, 0);
                    }
            }
        ST_T(4);                                               // GEMM2
    }
    // ---------------------------------------------------------------- write the item's rows back (rows beyond mul_k: the trash row)
#pragma unroll
    for (int rtp = 0; rtp < RTO; ++rtp)
        if (rtp < rto) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* __restrict__ t = tbase + rtab[16 * rtp + 4 * g + r];
#pragma unroll
                for (int c = 0; c < NC; ++c) t[ST_COL(c) * 16] = acc[rtp][c][r];
            }
        }
    ST_T(5);                                                   // write-back
}
#undef ST_COL

// dispatch code of an op (plan.py:st_schedule): TP items odd * 64 + x4 * 32 + MM * 4 + {RTO 1: 0, 2: 1, 4: 2}; Linear items 128 + x4 * 8 + MM
#ifdef ST_ONLY                // register-pressure audit of ONE instantiation (compile-only experiment)
#define ST_CASE(MMv, RTOv, rc, ODDv, X4v) case (ODDv * 64 + X4v * 32 + MMv * 4 + rc): if (ODDv * 64 + X4v * 32 + MMv * 4 + rc == ST_ONLY) item_st<MMv, RTOv, ODDv != 0, X4v != 0>(A, op, lds, lane, st, oi + 1 == o1); break;
#define ST_CASE_LIN(MMv, X4v) case (128 + X4v * 8 + MMv): if (128 + X4v * 8 + MMv == ST_ONLY) item_lin<MMv, X4v != 0>(A, op, lds, lane, st); break;
#else
#define ST_CASE(MMv, RTOv, rc, ODDv, X4v) case (ODDv * 64 + X4v * 32 + MMv * 4 + rc): item_st<MMv, RTOv, ODDv != 0, X4v != 0>(A, op, lds, lane, st, oi + 1 == o1); break;
#define ST_CASE_LIN(MMv, X4v) case (128 + X4v * 8 + MMv): item_lin<MMv, X4v != 0>(A, op, lds, lane, st); break;
#endif

__global__ __launch_bounds__(IS_NT, IS_NW / 2) void tp_st_kernel(const IsArgs A, const int* __restrict__ g_segs, const int* __restrict__ g_blocks,
                                                                 const int* __restrict__ g_phases, const int* __restrict__ g_groups,
                                                                 const int* __restrict__ g_ops, const float* __restrict__ g_stream,
                                                                 const int* __restrict__ g_rowtab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g = lane >> 4;
    const int64_t e0 = (int64_t)blockIdx.x * 16 + (lane & 15);
    const int64_t erow = e0 < A.rows ? e0 : A.rows - 1;
    float* __restrict__ stage = lds + A.stage_off;

    int* __restrict__ ctr = reinterpret_cast<int*>(lds + A.ctr_off);
    StState st;
#ifdef HG_PROF
    for (int k = 0; k < 12; ++k) st.t[k] = 0;
    st.last = __builtin_readcyclecounter();
    const unsigned long long t_begin = st.last;
#endif
#define s st

    for (int i = threadIdx.x; i < A.rowtab_off; i += IS_NT) lds[i] = 0.f;            // all segment tiles + the trash row
    {
        int* __restrict__ rt_l = reinterpret_cast<int*>(lds + A.rowtab_off);
        const int* __restrict__ rt_g = g_rowtab + A.rowtab_begin;
        for (int i = threadIdx.x; i < A.rowtab_len; i += IS_NT) rt_l[i] = rt_g[i];
    }

    for (int ph = 0; ph < A.nphase; ++ph) {
        const int* __restrict__ P = g_phases + ph * 8;
        const int b0 = P[0], b1 = P[1], g0 = P[2], g1 = P[3];
        int64_t er = erow;                                     // opaque per phase: the row-derived pointers of the staging code (seven Wigner
        asm volatile("" : "+v"(er));                           // blocks, source rows) are recomputed here instead of living through the items
        {   // hidden rows of the phase's radial weight generator (P[4]: 0 node, 1 edge branch; a phase never mixes them): under the staging
            const float* __restrict__ hrow = (P[4] ? A.h2[1] : A.h2[0]) + er * A.hidden + 4 * g;
#pragma unroll
            for (int G = 0; G < 4; ++G) st.hb[G] = *reinterpret_cast<const f32x4*>(hrow + 16 * G);
        }
        ST_T(6);                                               // zero fill / phase bookkeeping
        __syncthreads();                                       // every wave is done with the previous blocks (and the zero fill)
        ST_T(7);                                               // waiting for the slowest wave of the previous phase
        if (threadIdx.x == 0) *ctr = g0;
#pragma unroll 1
        for (int b = b0; b < b1; ++b) {
#ifdef HG_ABL_NOSTAGE
            if (A.rows > 0) continue;
#endif
            const int* __restrict__ B = g_blocks + b * 8;
            switch (B[4]) {
                case 0: stage_block<0>(A, B, stage, er, wave, lane); break;
                case 1: stage_block<1>(A, B, stage, er, wave, lane); break;
                case 2: stage_block<2>(A, B, stage, er, wave, lane); break;
                case 3: stage_block<3>(A, B, stage, er, wave, lane); break;
                case 4: stage_block<4>(A, B, stage, er, wave, lane); break;
                case 5: stage_block<5>(A, B, stage, er, wave, lane); break;
                case 6: stage_block<6>(A, B, stage, er, wave, lane); break;
                default: break;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ST_T(8);                                               // staging
        // work groups = all items of one (phase, output segment key), claimed largest first: dynamic balance, and a tile is only ever
        // updated by one wave between two barriers.  A group's weights are three contiguous streams (plan.py:st_schedule)
        while (true) {
            int gi = 0;
            if (lane == 0) gi = atomicAdd(ctr, 1);
            gi = __builtin_amdgcn_readfirstlane(gi);
            if (gi >= g1) break;
            const int* __restrict__ GR = g_groups + gi * 8;
            st.pa = reinterpret_cast<const f32x4*>(g_stream + GR[2]) + lane;
            st.pr = reinterpret_cast<const f32x4*>(g_stream + GR[3]) + lane;
            st.pc = reinterpret_cast<const f32x4*>(g_stream + GR[4]) + g;
#pragma unroll
            for (int G = 0; G < 4; ++G) st.wr[G] = st.pr[G * 64];      // radial fragments of the group's first row tile,
            st.pr += 256;
#if ST_RADAHEAD
            st.S = st_radial(st);                                      // its scale; the second row tile's fragments are requested
#endif
            const int o0 = GR[0], o1 = GR[1];
#ifdef HG_ABL_NOITEMS             // ablation: the launch without its items (wrong results; what the skeleton around them costs)
            if (A.rows > 0) continue;
#endif
#pragma unroll 1
            for (int oi = o0; oi < o1; ++oi) {
                const int* __restrict__ op = g_ops + oi * ST_OP_I32;
                switch (op[0]) {
                    ST_CASE(0, 1, 0, 0, 0) ST_CASE(0, 2, 1, 0, 0) ST_CASE(0, 4, 2, 0, 0) ST_CASE(0, 1, 0, 0, 1) ST_CASE(0, 2, 1, 0, 1) ST_CASE(0, 4, 2, 0, 1)
                    ST_CASE(1, 1, 0, 0, 0) ST_CASE(1, 2, 1, 0, 0) ST_CASE(1, 4, 2, 0, 0) ST_CASE(1, 1, 0, 0, 1) ST_CASE(1, 2, 1, 0, 1) ST_CASE(1, 4, 2, 0, 1)
                    ST_CASE(1, 1, 0, 1, 0) ST_CASE(1, 2, 1, 1, 0) ST_CASE(1, 4, 2, 1, 0) ST_CASE(1, 1, 0, 1, 1) ST_CASE(1, 2, 1, 1, 1) ST_CASE(1, 4, 2, 1, 1)
                    ST_CASE(2, 1, 0, 0, 0) ST_CASE(2, 2, 1, 0, 0) ST_CASE(2, 1, 0, 1, 0) ST_CASE(2, 2, 1, 1, 0)
                    ST_CASE(3, 1, 0, 0, 0) ST_CASE(3, 2, 1, 0, 0) ST_CASE(3, 1, 0, 1, 0) ST_CASE(3, 2, 1, 1, 0)
                    ST_CASE(4, 1, 0, 0, 0) ST_CASE(4, 1, 0, 1, 0)
                    ST_CASE(5, 1, 0, 0, 0) ST_CASE(5, 1, 0, 1, 0)
                    ST_CASE(6, 1, 0, 0, 0) ST_CASE(6, 1, 0, 1, 0)
                    ST_CASE_LIN(0, 0) ST_CASE_LIN(1, 0) ST_CASE_LIN(2, 0) ST_CASE_LIN(3, 0) ST_CASE_LIN(4, 0) ST_CASE_LIN(5, 0) ST_CASE_LIN(6, 0)
                    ST_CASE_LIN(0, 1) ST_CASE_LIN(1, 1)
                    default: break;
                }
            }
        }
    }

    ST_T(6);
    // ---------------------------------------------------------------- epilogue (as csrc/tp_is.hip): all four waves on one segment at a time
    // (edge index / validity recomputed from an opaque lane id: cold values are not carried through the item loop)
    int lane_e = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_e));
    const int64_t e = (int64_t)blockIdx.x * 16 + (lane_e & 15);
    const bool valid = e < A.rows;
    const int64_t erow_e = valid ? e : A.rows - 1;
    const int g_e = lane_e >> 4;
    for (int sg = 0; sg < A.nseg; ++sg) {
#ifdef HG_ABL_NOEPI
        if (A.rows > 0) continue;
#endif
        const int* __restrict__ S8 = g_segs + sg * 8;
        const int lk = S8[0], mul_k = S8[1], out_off = S8[3], out_mulp = S8[4], tile_off = S8[5], woff = S8[6], flags = S8[7];
        if (sg == 0 || (flags & SEG_NEWBATCH)) {
            __syncthreads();                                   // tiles complete / previous batch no longer read
            if (flags & SEG_NEWBATCH) {
                int lprev = -1;
                for (int s2 = sg; s2 < A.nseg; ++s2) {
                    const int* __restrict__ T8 = g_segs + s2 * 8;
                    if (s2 > sg && (T8[7] & SEG_NEWBATCH)) break;
                    const int l2 = T8[0];
                    if (!(T8[7] & SEG_UNROTATE) || l2 == lprev) continue;
                    lprev = l2;
                    const int nn = (2 * l2 + 1) * (2 * l2 + 1);
                    const float* __restrict__ D = A.wig + erow_e * A.nW + is_pick_wig_off(A, l2);
                    const int nj = (nn + 3) >> 2;
#pragma unroll 1
                    for (int j = wave; j < nj; j += IS_NW) {
                        int idx = 4 * j + g_e;
                        idx = idx < nn ? idx : nn - 1;
                        is_dma4(D + idx, stage + T8[6] + j * 64);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        const float* __restrict__ tile = lds + tile_off;
        const float* __restrict__ dst = stage + woff;
        switch (lk) {
            case 0: epilogue_is<0>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane_e); break;
            case 1: epilogue_is<1>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane_e); break;
            case 2: epilogue_is<2>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane_e); break;
            case 3: epilogue_is<3>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane_e); break;
            case 4: epilogue_is<4>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane_e); break;
            case 5: epilogue_is<5>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane_e); break;
            case 6: epilogue_is<6>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane_e); break;
            default: break;
        }
    }
    ST_T(9);                                                   // epilogue
#ifdef HG_PROF
    if (lane == 0) {
        for (int k = 0; k < 12; ++k) atomicAdd(&hg_prof_st_acc[k], st.t[k]);
        atomicAdd(&hg_prof_st_acc[15], st.last - t_begin);
    }
#endif
#undef s
}

#ifdef HG_PROF
extern "C" int hg_prof_st_read(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(hg_prof_st_acc), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(hg_prof_st_acc), z, sizeof(z));
    }
    return 0;
}
#endif

extern "C" int hg_tp_st(const float* const* src, const int64_t* src_stride, int nsrc, const float* h2_node, const float* h2_edge,
                        int hidden, const float* wig, int nW, const int32_t* wig_off, const float* stream, const int32_t* seg_table,
                        const int32_t* block_table, const int32_t* phase_table, const int32_t* group_table, const int32_t* op_table,
                        const int32_t* part_host, const int32_t* row_table, int lds_bytes,
                        const int64_t* const* src_idx, int rot_mask, float* out, int64_t out_stride, int64_t rows, void* stream_h) {
    HgDeviceGuard dev_guard(stream_h);
    if (rows <= 0) return 0;
    if (nsrc < 1 || nsrc > 4) return hg_fail(-2, "hg_tp_st: nsrc must be 1..4");
    if (hidden != 64) return hg_fail(-2, "hg_tp_st: the (padded) radial hidden width must be 64");
    if (!h2_node || !h2_edge) return hg_fail(-2, "hg_tp_st: both radial hidden row tensors are required");
    if (lds_bytes <= 0 || lds_bytes > 160 * 1024) return hg_fail(-2, "hg_tp_st: bad LDS size");
    if (!part_host || !row_table || !stream || !op_table || !group_table) return hg_fail(-2, "hg_tp_st: missing table");
    IsArgs A;
    for (int i = 0; i < 4; ++i) {
        A.src[i] = i < nsrc ? src[i] : src[0];
        A.sstride[i] = i < nsrc ? src_stride[i] : src_stride[0];
    }
    A.h2[0] = h2_node;
    A.h2[1] = h2_edge;
    A.hidden = hidden;
    A.wig = wig;
    A.nW = nW;
    for (int i = 0; i < 8; ++i) A.wig_off[i] = wig_off ? wig_off[i] : 0;
    A.out = out;
    A.ostride = out_stride;
    A.rows = rows;
    A.tile_shift = 0;
    const int32_t* p0 = part_host;                             // the single part of plan.IsSchedule.part_table
    A.nseg = p0[1], A.nphase = p0[3], A.trash_off = p0[4], A.stage_off = p0[5], A.ctr_off = p0[6];
    A.rowtab_off = p0[8], A.rowtab_begin = p0[9], A.rowtab_len = p0[10];
    if (p0[6] < p0[5] || p0[5] < p0[8] + p0[10] || p0[8] < p0[4] || lds_bytes < 4 * (p0[6] + 1)) return hg_fail(-2, "hg_tp_st: bad LDS layout");
    for (int i = 0; i < 4; ++i) A.idx[i] = (src_idx && i < nsrc) ? src_idx[i] : nullptr;
    A.rot_mask = rot_mask;
    if (rot_mask && !wig) return hg_fail(-2, "hg_tp_st: rotated sources need the Wigner rows");
    static unsigned char lds_attr_done[HG_MAX_DEVICES];
    if (int rc = hg_lds_attr_once(lds_attr_done, dev_guard.dev, (const void*)tp_st_kernel, 160 * 1024)) return rc;
    const unsigned grid = (unsigned)((rows + 15) / 16);
    hipLaunchKernelGGL(tp_st_kernel, dim3(grid), dim3(IS_NT), lds_bytes, (hipStream_t)stream_h, A, seg_table, block_table, phase_table, group_table,
                       op_table, stream, row_table);
    return hg_check_launch("hg_tp_st");
}

// tp_is.hip -- input-stationary variant of the fused equivariant edge kernel (gfx950, CDNA4).  Hand-written HIP.
//
// Same items and the same per-item arithmetic as tp_fused.hip (GEMM1 -> radial scale -> GEMM2 chained through MFMA C
// fragments), but a different loop nest and work split, driven by the measured HBM traffic of the segment-stationary kernel
// (profiles/r01c_tp_fused_hbm_pmc.md: 12x the algorithmic bytes, because every item re-stages its input-irrep span and the
// reuse distance is far beyond L2 / Infinity Cache):
//   * a workgroup = 4 waves = ONE tile of 16 edges; the tiles of ALL output segments live in LDS at once (<= 80 KB: 2 WG/CU);
//   * outer loop over the input irrep blocks ("phases", hamgnn_amd/plan.py:is_schedule): the block of the 16 edges (both
//     sources of the node branch) is staged ONCE by LDS-DMA, cooperatively, and every item that reads it runs in that phase,
//     spread over the four waves (the planner keeps all items of one (phase, output segment) on one wave, so no two waves
//     update the same tile between two barriers);
//   * B operands are read by all waves from the shared staged block; each input row is fetched once per launch (10.8 KB/edge
//     instead of ~170 KB/edge), and the exposed span latencies drop from one per item to one per phase.
// Items, fragments and weights are exactly those of the segment-stationary program (same planner output, regrouped).
#include "tp_stage.h"

// phase profiler (HG_PROF builds only, tests/bench_tp.py): per-wave shader-clock time between probes, summed over waves
#ifdef HG_PROF
__device__ unsigned long long hg_prof_is_acc[16];
struct ProfIs { unsigned long long t[12]; unsigned long long last; };
#define IS_PROF_ARG , ProfIs& prof
#define IS_PROF_PASS , prof
#define IS_T(k)                                                      \
    do {                                                             \
        const unsigned long long t_ = __builtin_readcyclecounter();  \
        prof.t[k] += t_ - prof.last;                                 \
        prof.last = t_;                                              \
    } while (0)
#else
#define IS_PROF_ARG
#define IS_PROF_PASS
#define IS_T(k)
#endif
// broadcast of lane q of every row of 16 lanes to that row: DPP row_newbcast (gfx90a+): lanes (g, *) <- lane (g, q).
// x * (lane q of v's row of 16 lanes) in ONE VALU instruction: v_mul_f32 with the DPP control on its first source (the compiler keeps the
// broadcast as a separate v_mov_b32_dpp: 2 000 of them in the default instantiation)
#define IS_MB_CASE(Q) case Q: asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "v"(x)); break;
__device__ __forceinline__ float is_mul_bcast(float v, float x, int q) {
    float o;
    switch (q) {
        IS_MB_CASE(0) IS_MB_CASE(1) IS_MB_CASE(2) IS_MB_CASE(3) IS_MB_CASE(4) IS_MB_CASE(5) IS_MB_CASE(6) IS_MB_CASE(7)
        IS_MB_CASE(8) IS_MB_CASE(9) IS_MB_CASE(10) IS_MB_CASE(11) IS_MB_CASE(12) IS_MB_CASE(13) IS_MB_CASE(14)
        default: asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "v"(x)); break;
    }
    return o;
}
#undef IS_MB_CASE

// One item on the workgroup's 16 edges.  stage: the phase's staged input block(s), image offset(piece p, row e) = 64 p + 4 e per
// source (pieces of the FULL irrep block: component a, channel piece s -> p = a * P1 + s).
// ODD: a super-path with l_i + l_sh + l_k odd (item[7], the reversed-column flag).  Its aligned-frame coupling is antisymmetric in m, so
// the coefficient of the centre column (m = 0) vanishes for every path of the item (checked by the planner's emulator): the kernel
// works on the 2 MM remaining columns only -- column slot c < MM is real column c, slot c >= MM is real column c + 1.
template <int MM, int RTM, bool SPLIT, bool ODD>
__device__ __forceinline__ void item_is(const IsArgs& A, const float* __restrict__ Wb, const int* __restrict__ it,
                                        float* __restrict__ lds, int64_t erow, int lane, const IsHidden& hbr, int hbr_cls IS_PROF_ARG) {
    constexpr int NCR = 2 * MM + 1;                            // real columns (fragment layouts of cf, tile columns)
    constexpr int NC = ODD ? 2 * MM : NCR;                     // column slots this item computes
#define IS_COL(c) ((ODD && (c) >= MM) ? (c) + 1 : (c))
// K-steps of GEMM2 beyond the item's rows are not issued; only the LAST row tile can hold such K-steps (rtm = ceil(rows / 16), plan._add_item)
#define IS_NK2_OK(rt, r) ((rt) + 1 < RTM || 4 * (rt) + (r) < nk2)
    constexpr int CW = NC > 7 ? (NC + 1) / 2 : NC;             // GEMM2 column chunk (keeps its accumulators <= 28 VGPRs)
    const int typ = it[0], so0 = it[1], so1 = it[2], in_mulp = it[4], li = it[5], neg = it[7];     // [1], [2]: stage offsets of the sources
    const int ksteps = it[8], mlp = it[10], x4 = it[17], nk2 = it[18];
    const int rto = it[22];                                     // 16-row tiles of GEMM2's output: the item's segment, or all members of a merged item
    const int g = lane >> 4, el = lane & 15;
    // GEMM2's output rows are addressed through the row table: entry = LDS offset of the row's centre column (m = 0) in its segment tile,
    // rows beyond the segment's multiplicity (fragment padding) point at the trash row -- no compare / select / multiply per row
    const int* __restrict__ rtab = reinterpret_cast<const int*>(lds + A.rowtab_off) + it[23];
    float* __restrict__ tbase = lds + (SPLIT ? A.tile_shift : 0) + (el - MM * 16);
    const float* __restrict__ stage = lds + A.stage_off;
    IS_T(0);                                                    // dispatch

    const int nsrc = so1 >= 0 ? 2 : 1;
    const int ngrp = (ksteps + 3) >> 2;
    const f32x4* __restrict__ aw = reinterpret_cast<const f32x4*>(Wb + it[11]) + lane;      // [src][G][rt][lane]
    f32x4 av_n[RTM];
    // CG coefficients cf[row, column] in PACKED form (plan._cf_block): one float4 per lane covers 16 (row tile, column) pairs -- lane (g, p) holds
    // the four rows 4 g + r of pair p -- requested with the radial operands, so nothing is left to wait for at the scale step, which broadcasts
    // pair p along the 16 lanes of row g by DPP (r3: one float4 load and four registers per PAIR, requested after GEMM1 and waited for at once)
    constexpr int NPAIR = RTM * NCR, NJ = (NPAIR + 15) / 16;
    f32x4 cfv[NJ];
    if (typ == 0) {
        const f32x4* __restrict__ cfp = reinterpret_cast<const f32x4*>(Wb + it[13] + NPAIR * 16) + lane;
#pragma unroll
        for (int j = 0; j < NJ; ++j) cfv[j] = cfp[j * 64];
    }
    // ---------------------------------------------------------------- radial scale s_e = W3^T h2 first (see tp_fused.hip)
    f32x4 S[RTM];
    if (typ == 0) {
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) S[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ hrow = (mlp ? A.h2[1] : A.h2[0]) + erow * A.hidden + 4 * g;
        const f32x4* __restrict__ w3 = reinterpret_cast<const f32x4*>(Wb + it[12]) + lane;
        const int hgrp = A.hidden >> 4;
        if (mlp == hbr_cls) {
            // the 16 edges' hidden rows of the phase's radial MLP (the B operands of these MFMAs, the same for every item of a branch) are
            // resident in registers: read once per phase by the kernel, under the staging -- the planner keeps a phase on one generator where
            // that is free (plan.is_schedule(separate_mlp)); r3 loaded them per item: 4 of an item's ~35 vector loads, and a second round
            // trip when they missed the L1.
            // r6: on the HALF-PRECISION matrix pipe with split operands (plan/program.py: w3_split_fill): x 2^s = hi + 2^-11 lo, hi = f16(x 2^s),
            // lo = f16((x 2^s - hi) 2^11);  S 2^(sw + sh) = W_hi h_hi + 2^-11 (W_hi h_lo + W_lo h_hi) -- 3 MFMAs of K = 32 per half of the 64 hidden units
            // = 6 x 16 pipe cycles per row tile where the fp32 form takes 16 x 32 (27.5 % of the launch's MFMAs), fp32 accumulation in two chains (the
            // half-precision MFMAs flush subnormal inputs: hence the scaled remainders), 22-bit operands (H moves by 6e-7 relative: profiles/r06_tp_is.md).
            // The weights' split twin follows the fp32 block: [t][rt][hi, lo][lane] float4 = 8 halves in the K-slot order of the resident rows.
            const f32x4* __restrict__ w3s = w3 + 4 * RTM * 64;
            f32x4 wh[2][RTM], wl[2][RTM], S1[RTM];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) {
                    wh[t][rt] = w3s[((t * RTM + rt) * 2) * 64];
                    wl[t][rt] = w3s[((t * RTM + rt) * 2 + 1) * 64];
                }
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) S1[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // all fragments requested and waited for before the first MFMA: the faster form (6.15 vs 6.55 ms per 131 072 edges, profiles/r06_tp_is.md section 3)
#ifndef K_S_NOBATCH               /* A/B builds only: the compiler's own placement of the waits */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) S[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[t][rt]), hbr.hi[t], S[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) S1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[t][rt]), hbr.lo[t], S1[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) S1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[t][rt]), hbr.hi[t], S1[rt], 0, 0, 0);
            }
#ifndef K_S_NOBATCH
            __builtin_amdgcn_sched_barrier(0);
#endif
            const float c0 = A.s_scale, c1 = A.s_scale * (1.f / 2048.f);
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) S[rt] = S[rt] * c0 + S1[rt] * c1;
        } else
#pragma unroll 1
        for (int G0 = 0; G0 < hgrp; G0 += 4) {
            f32x4 hb[4], wv[4][RTM];
#pragma unroll
            for (int G = 0; G < 4; ++G)
                if (G0 + G < hgrp) {
                    hb[G] = *reinterpret_cast<const f32x4*>(hrow + 16 * (G0 + G));
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt) wv[G][rt] = w3[((G0 + G) * RTM + rt) * 64];
                }
#pragma unroll
            for (int G = 0; G < 4; ++G)
                if (G0 + G < hgrp) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int rt = 0; rt < RTM; ++rt) S[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[G][rt][q], hb[G][q], S[rt], 0, 0, 0);
                }
        }
    }

    IS_T(1);                                                    // radial scale
#if defined(K_XDL_DUMMY) || defined(K_SMFMA_DUMMY)
    // A/B builds only (profiles/r06_tp_is.md section 8): MFMAs whose result nobody reads, on operands that have nothing to do with the item -- 6 per row tile in two chains,
    // where the round's half-precision experiment had its radial scale.  K_XDL_DUMMY: v_mfma_f32_16x16x32_f16 (K_XD_OP: another opcode of that pipe); K_SMFMA_DUMMY: the fp32
    // kind (control).  The launch's result must not change by a bit.  K_XD_DRAIN: no memory operation of the wave in flight around them; K_XD_ONE: one instead of 6 per row
    // tile; K_XD_NOPS: 64 idle cycles behind them; K_XD_ROLE: only the workgroups with bit 8 of their index set issue them (workgroups j and j + 256 share a CU).
#ifdef K_XD_ROLE
    if (typ == 0 && ((blockIdx.x >> 8) & 1)) {
#else
    if (typ == 0) {
#endif
        f32x4 dA[RTM], dB[RTM];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) dA[rt] = dB[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef K_XD_DRAIN
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef K_XDL_DUMMY
#if !defined(K_XD_OP) || K_XD_OP == 0
        typedef _Float16 k_vec __attribute__((ext_vector_type(8)));
#define K_XD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define K_XD_VAL(x) (_Float16)(x)
#define K_XD_N 8
#elif K_XD_OP == 1
        typedef __bf16 k_vec __attribute__((ext_vector_type(8)));
#define K_XD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define K_XD_VAL(x) (__bf16)(x)
#define K_XD_N 8
#else
        typedef _Float16 k_vec __attribute__((ext_vector_type(4)));
#define K_XD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0)
#define K_XD_VAL(x) (_Float16)(x)
#define K_XD_N 4
#endif
        k_vec xa, xb;
#pragma unroll
        for (int s_ = 0; s_ < K_XD_N; ++s_) { xa[s_] = K_XD_VAL(0.01f * (float)((lane + 3 * s_) & 15) - 0.07f); xb[s_] = K_XD_VAL(0.02f * (float)((lane * 3 + s_) & 7) - 0.06f); }
#if defined(K_XD_ONE)
        dA[0] = K_XD_MFMA(xa, xb, dA[0]);
#elif defined(K_XD_TWO)                                         /* two per row tile, independent */
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) dA[rt] = K_XD_MFMA(xa, xb, dA[rt]);
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) dB[rt] = K_XD_MFMA(xb, xa, dB[rt]);
#elif defined(K_XD_INDEP6)                                      /* 6 per row tile, every one on its own zero accumulator, summed on the VALU */
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) {
            const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 m0 = K_XD_MFMA(xa, xb, z4), m1 = K_XD_MFMA(xb, xa, z4), m2 = K_XD_MFMA(xa, xa, z4), m3 = K_XD_MFMA(xb, xb, z4), m4 = K_XD_MFMA(xa, xa + xb, z4), m5 = K_XD_MFMA(xa + xb, xb, z4);
            dA[rt] = m0 + m1;
            dB[rt] = (m2 + m3) + (m4 + m5);
        }
#elif defined(K_XD_CHAIN)                                       /* 6 in ONE back-to-back dependent chain */
#pragma unroll
        for (int t = 0; t < 6; ++t) dA[0] = K_XD_MFMA(xa, xb, dA[0]);
#elif defined(K_XD_SIX)                                         /* 6 per item, independent pairs at distance 2 */
#pragma unroll
        for (int t = 0; t < 3; ++t) { dA[0] = K_XD_MFMA(xa, xb, dA[0]); dB[0] = K_XD_MFMA(xb, xa, dB[0]); }
#else
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) dA[rt] = K_XD_MFMA(xa, xb, dA[rt]);
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) dB[rt] = K_XD_MFMA(xb, xa, dB[rt]);
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) dB[rt] = K_XD_MFMA(xa, xa, dB[rt]);
        }
#endif
#else
        const float xa = 0.01f * (float)(lane & 15) - 0.07f, xb = 0.02f * (float)(lane & 7) - 0.06f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) dA[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, xb, dA[rt], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) dB[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb, xa, dB[rt], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) dB[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, xa, dB[rt], 0, 0, 0);
        }
#endif
#if defined(K_XD_DRAIN) || defined(K_XD_NOPS)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#ifdef K_XD_LONG                                                /* K_XD_LONG x 64 idle cycles more */
#pragma unroll 1
        for (int w_ = 0; w_ < K_XD_LONG; ++w_) asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) asm volatile("" :: "v"(dA[rt]), "v"(dB[rt]));
#if defined(K_XD_DRAIN)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#endif
    // ---------------------------------------------------------------- GEMM1: mid = A1 fragments x staged block
    f32x4 mid[RTM][NC];
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
        for (int c = 0; c < NC; ++c) mid[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int P1 = in_mulp >> 2;                               // float4 pieces per component
    const int cdir = neg ? -P1 : P1;                           // column c -> component (neg ? a_hi - c : a_lo + c)
    const int c0p = (li - MM) * P1 + (neg ? (NCR - 1) * P1 : 0);
    // ONE loop over (source, K group): as a loop nest with two alternative inner loops the accumulators came back as 64 + register copies per
    // source at the loops' joins (ISA audit r4: 2 100 v_mov_b64 in the default instantiation).  The B operand of (column c, K group G) is read
    // through a running pointer per column (+ one K group per iteration, a jump at the switch to the second source): the K-steps of a group
    // are immediate offsets -- computed from (c, G, q) every K-step the addresses cost ~17 scalar + 5 vector instructions per NC MFMAs
    const int ntot = nsrc * ngrp;
    const int src_jump = (so1 - so0) - ngrp * 256;             // floats, applied once when t reaches the second source
    if (NCR <= 3 && x4) {                                      // permuted K: fragment (c, G) = piece cbase + 4G + g of row el
        const float* __restrict__ pc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) pc[c] = stage + so0 + (c0p + g + IS_COL(c) * cdir) * 64 + el * 4;
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[rt * 64];              // (requested inside the branch: ahead of it, both loops got a copy)
#pragma unroll 1
        for (int t = 0; t < ntot; ++t) {
            if (t == ngrp) {
#pragma unroll
                for (int c = 0; c < NC; ++c) pc[c] += src_jump;
            }
            f32x4 av[RTM], bv[NC];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = av_n[rt];
            if (t + 1 < ntot) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[((t + 1) * RTM + rt) * 64];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                bv[c] = *reinterpret_cast<const f32x4*>(pc[c]);
                pc[c] += 256;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        mid[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][q], bv[c][q], mid[rt][c], 0, 0, 0);
        }
    } else {                                                   // natural K: element (c, 4 sl + g) = piece cbase + sl, component g
        const float* __restrict__ pc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) pc[c] = stage + so0 + (c0p + IS_COL(c) * cdir) * 64 + el * 4 + g;
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[rt * 64];
        int nq = ksteps;                                       // K-steps left in this source (>= 4 except in its tail group)
#pragma unroll 1
        for (int t = 0; t < ntot; ++t) {
            if (t == ngrp) {
                nq = ksteps;
#pragma unroll
                for (int c = 0; c < NC; ++c) pc[c] += src_jump;
            }
            f32x4 av[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = av_n[rt];
            if (t + 1 < ntot) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[((t + 1) * RTM + rt) * 64];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nq) {
                    float b[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) b[c] = pc[c][q * 64];
                    // the NC operand reads stay together ahead of the MFMAs: left alone, the scheduler of the one-row-tile instantiations read every
                    // operand into ONE register right before its MFMA -- an LDS round trip per MFMA (ISA audit: 334 of 4 096 static MFMAs)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            mid[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][q], b[c], mid[rt][c], 0, 0, 0);
                }
            }
            nq -= 4;
#pragma unroll
            for (int c = 0; c < NC; ++c) pc[c] += 256;
        }
    }

    IS_T(2);                                                    // GEMM1
    if (typ == 0) {
        const f32x4* __restrict__ a2 = reinterpret_cast<const f32x4*>(Wb + it[14]) + lane;
        const f32x4* __restrict__ cf = reinterpret_cast<const f32x4*>(Wb + it[13]) + g;     // [rt][c][g] float4
        f32x4 a2_n[RTM];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[rt * 64];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int p = rt * NCR + IS_COL(c);
                f32x4 t = mid[rt][c] * S[rt];
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = is_mul_bcast(cfv[p >> 4][r], t[r], p & 15);
                mid[rt][c] = t;
            }

        // ------------------------------------------------------------ GEMM2: tile[w'', m] += L' fragments x mid
        // rows beyond mul_k (fragment padding) go to the shared trash row: no divergent branches (their L' columns are zero)
        if (CW == NC) {
            // software-pipelined over the output row tiles: the tile values of step rtp+1 (the MFMA accumulator init) are read
            // while the MFMAs of step rtp execute -- the LDS latency was exposed once per step (GEMM2 ran at half the MFMA
            // density of GEMM1 in the phase profile)
            float* __restrict__ tnext[4];
            f32x4 acc_n[NC];
#pragma unroll
            for (int r = 0; r < 4; ++r) tnext[r] = tbase + rtab[4 * g + r];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_n[c][r] = tnext[r][IS_COL(c) * 16];
#pragma unroll 1
            for (int rtp = 0; rtp < rto; ++rtp) {
                f32x4 av[RTM], acc[NC];
                float* __restrict__ trow[4];
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av[rt] = a2_n[rt];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = acc_n[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) trow[r] = tnext[r];
                if (rtp + 1 < rto) {
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[((rtp + 1) * RTM + rt) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) tnext[r] = tbase + rtab[16 * (rtp + 1) + 4 * g + r];
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // (volatile: neighbouring columns off one base register would be merged into ds_read2_b32, whose register pairs
                            //  then have to be split into the accumulator vectors with v_mov -- each behind a wait for the read just issued)
                            acc_n[c][r] = *(volatile __attribute__((address_space(3))) float*)(tnext[r] + IS_COL(c) * 16);
                        }
                    __builtin_amdgcn_sched_barrier(0);         // ... and requested ahead of this step's MFMAs
                }
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (IS_NK2_OK(rt, r)) {                // trailing K-steps hold only padding rows: not issued
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][r], mid[rt][c][r], acc[c], 0, 0, 0);
                        }
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) trow[r][IS_COL(c) * 16] = acc[c][r];
            }
        } else {
#pragma unroll 1
        for (int rtp = 0; rtp < rto; ++rtp) {
            f32x4 av[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = a2_n[rt];
            if (rtp + 1 < rto) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[((rtp + 1) * RTM + rt) * 64];
            }
            float* __restrict__ trow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) trow[r] = tbase + rtab[16 * rtp + 4 * g + r];
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += CW) {
                f32x4 acc[CW];                                 // tile values are the accumulator init (C operand)
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c0 + c < NC) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[c][r] = trow[r][IS_COL(c0 + c) * 16];
                    }
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (IS_NK2_OK(rt, r)) {                // trailing K-steps hold only padding rows: not issued
#pragma unroll
                            for (int c = 0; c < CW; ++c)
                                if (c0 + c < NC)
                                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][r], mid[rt][c0 + c][r], acc[c], 0, 0, 0);
                        }
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c0 + c < NC) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) trow[r][IS_COL(c0 + c) * 16] = acc[c][r];
                    }
            }
        }
        }
    } else {
        // plain o3.Linear item (PairInteractionBlock skip): rows are output channels; add straight into the tile
        const int row0 = it[16];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
        {                                                      // a row tile's reads, then its writes (as "+=" per element the compiler chains
            float* __restrict__ t0[4];                         // read -> wait -> write through all rows: they may alias for all it knows)
            float told[4][NC];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t0[r] = tbase + rtab[row0 + 16 * rt + 4 * g + r];
#pragma unroll
                for (int c = 0; c < NC; ++c) told[r][c] = t0[r][IS_COL(c) * 16];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) t0[r][IS_COL(c) * 16] = told[r][c] + mid[rt][c][r];
        }
    }
    IS_T(3);                                                    // scale-mul + GEMM2 + write-back
#undef IS_COL
#undef IS_NK2_OK
}

// lite_mode items (message_passing.py:197-206: unweighted uvu product folded with its o3.Linear block), rows = output channels:
//   typ 2 (IT_LINC): tile[w, m] += cf[m] * sum_u A[u, w] x[u, src(m)]        one path (i, l_sh, k): one weight matrix, a coefficient per column
//   typ 4 (IT_LINM): tile[w, m] += sum_u A_m[u, w] x[u, src(m)]              ALL paths of (i, k) folded: A_m = sum_paths cf_path[m] A_path
// one column at a time (the accumulators of one column only), natural-K operands, the next fragment group requested under the MFMAs
template <int RTM>
__device__ __forceinline__ void item_lite(const IsArgs& A, const float* __restrict__ Wb, const int* __restrict__ it, float* __restrict__ lds, int lane) {
    const int typ = it[0], so0 = it[1], so1 = it[2], in_mulp = it[4], li = it[5], mm = it[6], neg = it[7], ksteps = it[8], x4 = it[17];
    const int g = lane >> 4, el = lane & 15;
    const int nc = 2 * mm + 1;
    const int* __restrict__ rtab = reinterpret_cast<const int*>(lds + A.rowtab_off) + it[23] + it[16];
    float* __restrict__ tbase = lds + A.tile_shift + (el - mm * 16);      // (split launches without a post-op: this wave's private tile copy)
    const float* __restrict__ stage = lds + A.stage_off;
    const int nsrc = so1 >= 0 ? 2 : 1;
    const int ngrp = (ksteps + 3) >> 2;
    const int P1 = in_mulp >> 2;
    const int cdir = neg ? -P1 : P1;
    const int c0p = (li - mm) * P1 + (neg ? (nc - 1) * P1 : 0);
    const int colstride = typ == 4 ? it[13] : 0;               // floats between the fragment sets of two columns
    int roff[RTM][4];
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) roff[rt][r] = rtab[16 * rt + 4 * g + r];
    const int nfr = nsrc * ngrp;
#pragma unroll 1
    for (int c = 0; c < nc; ++c) {
        const f32x4* __restrict__ aw = reinterpret_cast<const f32x4*>(Wb + it[11] + c * colstride) + lane;      // [src][G][rt][lane]
        const float cfc = typ == 2 ? Wb[it[13] + c] : 1.f;
        f32x4 acc[RTM], av_n[RTM];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) {
            acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            av_n[rt] = aw[rt * 64];
        }
#pragma unroll 1
        for (int f = 0; f < nfr; ++f) {
            const int si = f >= ngrp, G = f - si * ngrp;
            const float* __restrict__ fb = stage + (si ? so1 : so0) + (c0p + c * cdir + 4 * G) * 64 + el * 4 + g;
            f32x4 av[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = av_n[rt];
            if (f + 1 < nfr) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[((f + 1) * RTM + rt) * 64];
            }
            const int nq = ksteps - 4 * G;
            float b[4];
            if (x4) {                                          // permuted K (unfolded items whose channel block is a multiple of 16): one float4
                const f32x4 bv = *reinterpret_cast<const f32x4*>(stage + (si ? so1 : so0) + (c0p + c * cdir + 4 * G + g) * 64 + el * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = bv[q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = q < nq ? fb[q * 64] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][q], b[q], acc[rt], 0, 0, 0);
        }
        // all tile reads, then all writes: written as "+=" the compiler keeps every read behind the previous write (the rows may alias
        // for all it knows) -- one LDS round trip per ELEMENT (ISA audit, profiles/r03_lite.md).  Rows are distinct except the shared
        // trash row of padding rows, whose value nobody reads.
        float told[RTM][4];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) told[rt][r] = tbase[roff[rt][r] + c * 16];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) tbase[roff[rt][r] + c * 16] = told[rt][r] + cfc * acc[rt][r];
    }
}

// lite_mode STREAM (plan._lite_streams, r4): the folded items of one phase dealt to the waves as streams of UNIFORM steps.  A task = (segment, one
// 16-row tile, column m or pair +-m); a step = one fragment (16 rows x up to 16 input channels, natural K) + two descriptor words; only the
// K-steps that hold channels are issued (1..4 MFMAs, twice that for a pair): the input irreps with 2-12 channels fill a quarter or three
// quarters of a K group.  The accumulators of a task live in registers and are added into the tile at its last step through the row table.
// One instantiation for every row-tile count, so the request ring is SL_RING deep at 4 registers per slot; descriptors arrive in blocks of SL_RING steps by
// ONE scalar load a block ahead (the r3 runs -- one stream per (phase, segment, row chunk), rtm row tiles per step, profiles/r03_lite.md --
// carried a scalar load per step, which the step's MFMAs waited for: a wave can only wait for scalar loads with lgkmcnt(0)).
#ifndef SL_RING
#define SL_RING 4                // 4 (r4, with 8 waves per workgroup: streams of ~25 steps, less padding) or 8
#endif
typedef int i32xd __attribute__((ext_vector_type(2 * SL_RING)));          // the descriptors of SL_RING steps: one scalar load
__device__ __forceinline__ void stream_lite(const IsArgs& A, const float* __restrict__ Wb, const int* __restrict__ it, float* __restrict__ lds, int lane) {
    constexpr int RING = SL_RING;
    static_assert(RING == 8 || RING == 4, "descriptor blocks are 2 * RING dwords (plan.LITE_SRING): s_load_dwordx16 / x8");
    const int g = lane >> 4, el = lane & 15;
    const int nsteps = it[8];                                  // a multiple of RING (no-op steps at the end), RING more slots behind
    const int* __restrict__ rtab = reinterpret_cast<const int*>(lds + A.rowtab_off) + 4 * g;
    float* __restrict__ tbase = lds + A.tile_shift + (el - 256);          // column field = m + 16, row-table entries point at the centre column
    // B operand of K-step q: channel g of piece q, this lane's edge.  All four pieces behind the base are read whether the step issues their
    // K-steps or not (no clamp, no branch around the reads: what is not issued never reaches an accumulator)
    const char* __restrict__ stage = reinterpret_cast<const char*>(lds + A.stage_off + el * 4 + g);      // + (descriptor & 0x3ff00): piece index << 8 = byte offset
    const f32x4* __restrict__ aw = reinterpret_cast<const f32x4*>(Wb + it[11]) + lane;       // step t: aw[t * 64]
    const i32xd* __restrict__ dsc = reinterpret_cast<const i32xd*>(Wb + it[12]);              // uniform, 64-byte aligned: s_load_dwordx16 / x8
    f32x4 ring[RING];
#pragma unroll
    for (int j = 0; j < RING; ++j) {
        ring[j] = aw[j * 64];
        __builtin_amdgcn_sched_barrier(0);                     // slot order = request order (the scheduler issued the priming loads back to front otherwise, and the loop head's one static wait became vmcnt(0))
    }
    i32xd dc = dsc[0];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    int roff[4] = {0, 0, 0, 0};
    float bn[4], bnb[4];
    {
        const float* __restrict__ fb = reinterpret_cast<const float*>(stage + (dc[0] & 0x3ff00));
        const float* __restrict__ fc = reinterpret_cast<const float*>(stage + (dc[1] & 0x3ff00));
        const int sgn = dc[1] << 31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bn[q] = fb[q * 64];
            bnb[q] = __builtin_bit_cast(float, __builtin_bit_cast(int, fc[q * 64]) ^ sgn);
        }
    }
#pragma unroll 1
    for (int t0 = 0; t0 < nsteps; t0 += RING) {
        const i32xd dnx = dsc[t0 / RING + 1];
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const f32x4 av = ring[j];
            const int d = dc[2 * j], e1 = dc[2 * j + 1];
            float b[4], bb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                b[q] = bn[q];
                bb[q] = bnb[q];
            }
            {                                                  // operands of the next step, requested before this step's MFMAs (a wave issues in order)
                const int dn = j + 1 < RING ? dc[(2 * j + 2) % (2 * RING)] : dnx[0], en = j + 1 < RING ? dc[(2 * j + 3) % (2 * RING)] : dnx[1];
                const float* __restrict__ fb = reinterpret_cast<const float*>(stage + (dn & 0x3ff00));
                const float* __restrict__ fc = reinterpret_cast<const float*>(stage + (en & 0x3ff00));
                const int sgn = en << 31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bn[q] = fb[q * 64];
                    bnb[q] = __builtin_bit_cast(float, __builtin_bit_cast(int, fc[q * 64]) ^ sgn);
                }
            }
            if (d & 4) {                                       // first step of a task: its rows, fresh accumulators
                const int* __restrict__ rp = rtab + ((d >> 23) << 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) roff[r] = rp[r];
                acc = acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const int nq1 = d & 3;                             // K-steps - 1
            if (e1 < 0) {                                      // paired: both columns on this fragment, two independent accumulator chains
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b[0], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bb[0], acc2, 0, 0, 0);
                if (nq1 >= 1) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], b[1], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bb[1], acc2, 0, 0, 0);
                    if (nq1 >= 2) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], b[2], acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bb[2], acc2, 0, 0, 0);
                        if (nq1 >= 3) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], b[3], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bb[3], acc2, 0, 0, 0);
                        }
                    }
                }
            } else {                                           // single column: one chain (a conditionally updated second one comes back as
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b[0], acc, 0, 0, 0);      // register copies behind an MFMA-latency wait at the join)
                if (nq1 >= 1) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], b[1], acc, 0, 0, 0);
                    if (nq1 >= 2) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], b[2], acc, 0, 0, 0);
                        if (nq1 >= 3) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], b[3], acc, 0, 0, 0);
                    }
                }
            }
            if (d & 8) {                                       // last step of the task: add into the tile (all reads of a column, then its writes)
                const int tc = (d >> 14) & 0x1f0;              // (m + 16) * 16
                if (e1 < 0) {
                    const int tcb = (e1 >> 14) & 0x1f0;
                    float told[4], toldb[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        told[r] = tbase[roff[r] + tc];
                        toldb[r] = tbase[roff[r] + tcb];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tbase[roff[r] + tc] = told[r] + acc[r];
                        tbase[roff[r] + tcb] = toldb[r] + acc2[r];
                    }
                } else {
                    float told[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) told[r] = tbase[roff[r] + tc];
#pragma unroll
                    for (int r = 0; r < 4; ++r) tbase[roff[r] + tc] = told[r] + acc[r];
                }
            }
            ring[j] = aw[(t0 + j + RING) * 64];                // AFTER the step's MFMAs have read slot j: requested before them, the new value cannot share the slot's registers
                                                               // and the compiler rotates the whole ring with v_mov at the back-edge, waiting vmcnt(0) there
        }
        dc = dnx;
    }
}

// lite_mode segment post-op (message_passing.py:209-215: combine_messages = LinearScaleWithWeights on the summed branches), the last phase of a
// lite program's part: tile[v, m] <- sum_w Lc[w, v] * s_e[w] * tile[w, m] with s_e = W3^T h2 by MFMA, in place per chunk of four columns
// (every row of a column is read before one is written; the segment belongs to this wave alone in its phase).
template <int RTO>
__device__ __forceinline__ void post_is(const IsArgs& A, const float* __restrict__ Wb, const int* __restrict__ it, float* __restrict__ lds,
                                        int64_t erow, int lane) {
    asm volatile("" : "+v"(erow));                             // (per-edge addresses are formed HERE: hoisted to the kernel's entry they were spilled)
    const int g = lane >> 4, el = lane & 15;
    const int lk = it[20], mul_k = it[21];
    const int nco = 2 * lk + 1;
    const int* __restrict__ rtab = reinterpret_cast<const int*>(lds + A.rowtab_off) + it[23];
    float* __restrict__ tb = lds + (el - lk * 16);
    const float* __restrict__ hrow = A.h2[0] + erow * A.hidden + 4 * g;
    const f32x4* __restrict__ w3 = reinterpret_cast<const f32x4*>(Wb + it[12]) + lane;          // [G][rto][lane]
    const f32x4* __restrict__ a2 = reinterpret_cast<const f32x4*>(Wb + it[14]) + lane;          // [rtp][rt][lane]
    // all loop bounds are compile-time (RTO row tiles, hidden = 64): the loads of a stage are issued together (the run-time-bounded version
    // exposed one L2 round trip per fragment: ~35 us per 64-channel segment and 16 edges)
    f32x4 S[RTO];
#pragma unroll
    for (int rt = 0; rt < RTO; ++rt) S[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the segment's Lc fragments: resident for all its columns and requested ahead of the radial phase for one or two row tiles; for three or four
    // (64 registers; these segments have few columns) requested per output row tile inside the column loop -- the lite instantiation runs four
    // waves per SIMD (128 registers) since r4
    constexpr bool RES = RTO <= 2;
    f32x4 av[RES ? RTO : 1][RTO];
    if (RES) {
#pragma unroll
        for (int rtp = 0; rtp < RTO; ++rtp)
#pragma unroll
            for (int rt = 0; rt < RTO; ++rt) av[RES ? rtp : 0][rt] = a2[(rtp * RTO + rt) * 64];
    }
    const int hgrp = A.hidden >> 4;
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        if (G < hgrp) {
            const f32x4 hb = *reinterpret_cast<const f32x4*>(hrow + 16 * G);
            f32x4 wv[RTO];
#pragma unroll
            for (int rt = 0; rt < RTO; ++rt) wv[rt] = w3[(G * RTO + rt) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RTO; ++rt) S[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[rt][q], hb[q], S[rt], 0, 0, 0);
            if (RTO >= 3) __builtin_amdgcn_sched_barrier(0);   // (three or four row tiles: the requests of ONE hidden group in flight, not of all four)
        }
    }
    for (int G = 4; G < hgrp; ++G) {                           // (hidden > 64)
        const f32x4 hb = *reinterpret_cast<const f32x4*>(hrow + 16 * G);
#pragma unroll
        for (int rt = 0; rt < RTO; ++rt) {
            const f32x4 wv = w3[(G * RTO + rt) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) S[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[q], hb[q], S[rt], 0, 0, 0);
        }
    }
    int rowoff[RTO][4];                                         // rows beyond mul_k (fragment padding): the trash row
#pragma unroll
    for (int rt = 0; rt < RTO; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) rowoff[rt][r] = rtab[16 * rt + 4 * g + r];
    constexpr int CH = RTO == 1 ? 4 : (RTO == 2 ? 2 : 1);       // columns per chunk (register budget: 128 per wave)
#pragma unroll 1
    for (int c0 = 0; c0 < nco; c0 += CH) {
        f32x4 md[RTO][CH];
#pragma unroll
        for (int rt = 0; rt < RTO; ++rt)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int cc = c0 + c < nco ? c0 + c : c0;      // tail: recompute column c0 (dropped below)
#pragma unroll
                for (int r = 0; r < 4; ++r) md[rt][c][r] = (16 * rt + 4 * g + r < mul_k) ? tb[rowoff[rt][r] + cc * 16] * S[rt][r] : 0.f;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (RES) {
#pragma unroll
            for (int rtp = 0; rtp < RTO; ++rtp) {
                f32x4 acc[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rt = 0; rt < RTO; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rtp][rt][r], md[rt][c][r], acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c0 + c < nco) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) tb[rowoff[rtp][r] + (c0 + c) * 16] = acc[c][r];
                    }
            }
        } else {
#pragma unroll 1
            for (int rtp = 0; rtp < RTO; ++rtp) {              // (not unrolled: all RTO x RTO fragment requests at once were 64 registers)
                f32x4 avr[RTO], acc[CH];
                int ro[4];
#pragma unroll
                for (int rt = 0; rt < RTO; ++rt) avr[rt] = a2[(rtp * RTO + rt) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) ro[r] = rtab[16 * rtp + 4 * g + r];
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rt = 0; rt < RTO; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(avr[rt][r], md[rt][c][r], acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c0 + c < nco) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) tb[ro[r] + (c0 + c) * 16] = acc[c][r];
                    }
            }
        }
    }
}

#define IS_CASE(MMv, RTMv) \
    case (MMv * 8 + RTMv): item_is<MMv, RTMv, SPLIT, false>(A, g_W, it, lds, erow, lane, hbr, hbr_cls IS_PROF_PASS); break;
#define IS_CASE_ODD(MMv, RTMv) \
    case (64 + MMv * 8 + RTMv): item_is<MMv, RTMv, SPLIT, true>(A, g_W, it, lds, erow, lane, hbr, hbr_cls IS_PROF_PASS); break;

// A launch runs `nparts` sub-schedules (blockIdx.y) of the same program: every part owns a disjoint set of output segments and the
// phases / groups / items that feed them (plan.py:is_schedule(parts=...)).  One part = the whole program (large edge counts); several
// parts spread ONE 16-edge tile's serial 34 k-MFMA pass over several workgroups when there are fewer tiles than CUs (small crystals).
// part record, int32[16]: {first segment, segments, first phase, phases, trash_off, stage_off, ctr_off (float offsets in the LDS),
// copy_stride, rowtab_off, rowtab_begin, rowtab_len, lite, 0, 0, 0, 0}.  copy_stride > 0: each of the four waves accumulates
// into its own copy of the part's tiles (copy w at + w * copy_stride), so all waves can work on one output segment at once; the copies are
// summed before the epilogue.  r5: several parts may share a segment range and split its PHASES (plan.is_schedule ("2d", P, K); replayed hipGraphs only):
// their segments are flagged SEG_ATOMIC, the epilogues add into zero-filled rows.
#define IS_PART_I32 16

template <bool SPLIT, bool LITE>
__global__ __launch_bounds__(LITE ? 64 * IS_NW_LITE : IS_NT, (LITE ? IS_NW_LITE : IS_NW) / 2) void tp_is_kernel(const IsArgs A0, const int* __restrict__ g_segs, const int* __restrict__ g_blocks,
                                                       const int* __restrict__ g_phases, const int* __restrict__ g_groups,
                                                       const int* __restrict__ g_items, const float* __restrict__ g_W,
                                                       const int* __restrict__ g_parts, const int* __restrict__ g_rowtab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = LITE ? IS_NW_LITE : IS_NW, NT = 64 * NW; // lite programs: 8 waves on the tile (four per SIMD at <= 128 VGPRs; the default
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63; // instantiation needs 219)
    const int g = lane >> 4;
    // (an XCD-aware tile order -- contiguous tile ranges per XCD, workgroup b -> XCD b % 8 -- was measured on the 10 k-atom crystal:
    //  49.6 ms per launch with and without it, profiles/r02_tp_is_experiments.md: the gathered node rows are not what the waves wait for)
    const int64_t e = (int64_t)blockIdx.x * 16 + (lane & 15);
    const bool valid = e < A0.rows;
    const int64_t eslot = valid ? e : A0.rows - 1;
    const int64_t erow = A0.eperm ? A0.eperm[eslot] : eslot;    // the edge whose rows this slot reads (receiver-major launches: hamgnn_amd/topo.py)
    // SPLIT = false: one part = the whole program, every schedule scalar comes straight from the kernel arguments (the large-graph path,
    // identical to the single-schedule kernel); SPLIT = true: blockIdx.y selects the part, its scalars replace the arguments'
    const int* __restrict__ PT = g_parts + (SPLIT ? blockIdx.y : 0) * IS_PART_I32;
    IsArgs Asplit;
    if constexpr (SPLIT) {
        Asplit = A0;
        Asplit.trash_off = PT[4];
        Asplit.stage_off = PT[5];
        Asplit.ctr_off = PT[6];
        Asplit.rowtab_off = PT[8];
        Asplit.rowtab_begin = PT[9];
        Asplit.rowtab_len = PT[10];
        Asplit.tile_shift = wave * PT[7];
    }
    const IsArgs& A = SPLIT ? Asplit : A0;
    const int seg0 = SPLIT ? PT[0] : 0, seg1 = SPLIT ? PT[0] + PT[1] : A0.nseg, ph0 = SPLIT ? PT[2] : 0, ph1 = SPLIT ? PT[2] + PT[3] : A0.nphase;
    const int copy_stride = SPLIT ? PT[7] : 0;
    float* __restrict__ stage = lds + A.stage_off;
    int* __restrict__ ctr = reinterpret_cast<int*>(lds + A.ctr_off);
#ifdef HG_PROF
    ProfIs prof;
    for (int k = 0; k < 12; ++k) prof.t[k] = 0;
    prof.last = __builtin_readcyclecounter();
    const unsigned long long t_begin = prof.last;
#endif

    for (int i = threadIdx.x; i < A.rowtab_off; i += NT) lds[i] = 0.f;            // all segment tiles (all copies) + trash rows
    {
        int* __restrict__ rt_l = reinterpret_cast<int*>(lds + A.rowtab_off);
        const int* __restrict__ rt_g = g_rowtab + A.rowtab_begin;
        for (int i = threadIdx.x; i < A.rowtab_len; i += NT) rt_l[i] = rt_g[i];
    }
    IS_T(4);                                                   // zero fill

    for (int ph = ph0; ph < ph1; ++ph) {
        const int* __restrict__ P = g_phases + ph * 8;
        const int b0 = P[0], b1 = P[1], g0 = P[2], g1 = P[3];
        // the hidden rows of the phase's radial MLP (P[4]; -1: none / hidden width != 64), resident for its items (item_is): requested here,
        // first used after the staging
        const int hbr_cls = (A.hidden == 64 && A.s_split && (P[4] == 0 || (P[4] == 1 && A.h2[1]))) ? P[4] : -1;
        IsHidden hbr;                                          // split into (hi, lo) halves once per phase, in the K-slot order of the weights' twins
        {
            const float* __restrict__ hrow = (hbr_cls == 1 ? A.h2[1] : A.h2[0]) + erow * A.hidden + 4 * g;
            f32x4 hv[4];
#pragma unroll
            for (int G = 0; G < 4; ++G) hv[G] = hbr_cls >= 0 ? *reinterpret_cast<const f32x4*>(hrow + 16 * G) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_) {
                    const float x = hv[2 * t + (s_ >> 2)][s_ & 3] * IS_SPLIT_H_SCALE;
                    const _Float16 h_ = (_Float16)x;
                    hbr.hi[t][s_] = h_;
                    hbr.lo[t][s_] = (_Float16)((x - (float)h_) * 2048.f);
                }
        }
        __syncthreads();                                       // every wave is done with the previous blocks (and the zero fill)
        IS_T(5);                                               // waiting for the slowest wave of the previous phase
        if (threadIdx.x == 0) *ctr = g0;
#pragma unroll 1
        for (int b = b0; b < b1; ++b) {
            const int* __restrict__ B = g_blocks + b * 8;
            switch (B[4]) {
                case 0: stage_block<0, NW>(A, B, stage, erow, wave, lane); break;
                case 1: stage_block<1, NW>(A, B, stage, erow, wave, lane); break;
                case 2: stage_block<2, NW>(A, B, stage, erow, wave, lane); break;
                case 3: stage_block<3, NW>(A, B, stage, erow, wave, lane); break;
                case 4: stage_block<4, NW>(A, B, stage, erow, wave, lane); break;
                case 5: stage_block<5, NW>(A, B, stage, erow, wave, lane); break;
                case 6: stage_block<6, NW>(A, B, stage, erow, wave, lane); break;
                default: break;
            }
#ifdef HG_PROF                     // staging by kind: 8 = plain rows (LDS-DMA issue), 9 = gathered l = 0 rows, 10 = rotated l = 1..3, 11 = rotated l >= 4
            if (!((A.rot_mask >> B[0]) & 1)) { IS_T(8); } else if (B[4] == 0) { IS_T(9); } else if (B[4] <= 3) { IS_T(10); } else { IS_T(11); }
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        IS_T(6);                                               // staging the phase's input blocks
        // work groups = all items of one (phase, output segment), claimed largest-first: dynamic balance, and a tile is only ever
        // updated by one wave between two barriers (the order of the adds into a tile cell is the program's: phases, then a group's items).
        // Parts with private tile copies (split launches of small crystals): every item is its own group and the groups are DEALT, not claimed
        // (r6) -- group g0 + k * NW + w is the k-th of wave w (planner: LPT on its cost model) -- so the content of every copy, and with the
        // fixed fold below the launch's whole summation order, does not depend on which wave was faster: two forwards agree bit for bit
        int gi_dealt = g0 + __builtin_amdgcn_readfirstlane(wave);      // (scalar: as a vector register it spilled in the <SPLIT, LITE> instantiation)
        while (true) {
            int gi = 0;
#if defined(IS_DYNAMIC_CLAIM)                                   /* A/B builds only (profiles/r06_tp_is.md section 5): the dynamic claims of rounds 2-5 */
            if (false) {
#elif defined(IS_DEAL_ALL)                                      /* A/B builds only (section 7): single-part launches dealt round-robin as well -- the four waves start in lock step */
            if (true) {
#else
            if (SPLIT && copy_stride) {
#endif
                gi = gi_dealt;
                gi_dealt += NW;
            } else {
                if (lane == 0) gi = atomicAdd(ctr, 1);
                gi = __builtin_amdgcn_readfirstlane(gi);
            }
            if (gi >= g1) break;
            const int ib = g_groups[2 * gi], ie = g_groups[2 * gi + 1];
            for (int ii = ib; ii < ie; ++ii) {
                const int* __restrict__ it = g_items + ii * 24;
                if (LITE) {                                    // lite_mode programs: their own (small) items, their own kernel instantiation
                    IS_T(0);                                   // dispatch / claim
                    if (it[0] == 3) {                          // post-op of one segment (the part's last phase)
                        if (it[22] == 1) post_is<1>(A, g_W, it, lds, erow, lane);
                        else if (it[22] == 2) post_is<2>(A, g_W, it, lds, erow, lane);
                        else if (it[22] == 3) post_is<3>(A, g_W, it, lds, erow, lane);
                        else post_is<4>(A, g_W, it, lds, erow, lane);
                    }
                    else if (it[0] == 6) stream_lite(A, g_W, it, lds, lane);
                    else if (it[9] == 1) item_lite<1>(A, g_W, it, lds, lane);
                    else if (it[9] == 2) item_lite<2>(A, g_W, it, lds, lane);
                    else if (it[9] == 3) item_lite<3>(A, g_W, it, lds, lane);
                    else item_lite<4>(A, g_W, it, lds, lane);
                    if (it[0] == 3) { IS_T(3); } else { IS_T(2); }       // (profile slots: 2 = lite items / runs, 3 = post-ops)
                    continue;
                }
                switch (it[6] * 8 + it[9] + ((it[0] == 0 && it[7]) ? 64 : 0)) {
                    IS_CASE(0, 1) IS_CASE(0, 2) IS_CASE(0, 3) IS_CASE(0, 4)
                    IS_CASE(1, 1) IS_CASE(1, 2) IS_CASE(1, 3) IS_CASE(1, 4)
                    IS_CASE(2, 1) IS_CASE(2, 2) IS_CASE(2, 3)              // row-tile table of plan.py:rtm_max (4,4,3,2,2,1,1): the r1 shapes
                    IS_CASE(3, 1) IS_CASE(3, 2)                            // <2,4>, <3,3>, <5,2> held 80-88 accumulator VGPRs and spilled
                    IS_CASE(4, 1) IS_CASE(4, 2)
                    IS_CASE(5, 1)
                    IS_CASE(6, 1)
                    IS_CASE_ODD(1, 1) IS_CASE_ODD(1, 2) IS_CASE_ODD(1, 3) IS_CASE_ODD(1, 4)      // odd super-paths: centre column skipped
                    IS_CASE_ODD(2, 1) IS_CASE_ODD(2, 2) IS_CASE_ODD(2, 3)
                    IS_CASE_ODD(3, 1) IS_CASE_ODD(3, 2)
                    IS_CASE_ODD(4, 1) IS_CASE_ODD(4, 2)
                    IS_CASE_ODD(5, 1)
                    IS_CASE_ODD(6, 1)
                    default: break;
                }
            }
        }
    }

    if (SPLIT && copy_stride) {                                // split launch: fold the waves' private tile copies into copy 0
        __syncthreads();
        for (int i = threadIdx.x; i < copy_stride; i += NT) {
            float acc = lds[i];
#pragma unroll
            for (int k = 1; k < NW; ++k) acc += lds[i + k * copy_stride];
            lds[i] = acc;
        }
    }
    // ---------------------------------------------------------------- epilogue: all four waves on one segment at a time; the Wigner
    // blocks of a batch of segments (one block per l, as many l as fit the staging area) are staged together by LDS-DMA
    IsScan scan;                                               // fused scatter (tp_stage.h): the slots' runs of equal receivers
    scan.last = true, scan.row = 0;
    if (!SPLIT && A0.run_id) scan = is_scan_setup(valid ? A0.run_id[e] : -1 - (int)(lane & 15), lane & 15);
    for (int sg = seg0; sg < seg1; ++sg) {
        const int* __restrict__ S8 = g_segs + sg * 8;
        const int lk = S8[0], mul_k = S8[1], out_off = S8[3], out_mulp = S8[4], tile_off = S8[5], woff = S8[6], flags = S8[7];
        if (sg == seg0 || (flags & SEG_NEWBATCH)) {
            __syncthreads();                                   // tiles complete / previous batch no longer read
            if (flags & SEG_NEWBATCH) {
                int lprev = -1;
                for (int s2 = sg; s2 < seg1; ++s2) {
                    const int* __restrict__ T8 = g_segs + s2 * 8;
                    if (s2 > sg && (T8[7] & SEG_NEWBATCH)) break;
                    const int l2 = T8[0];
                    if (!(T8[7] & SEG_UNROTATE) || l2 == lprev) continue;      // one block per l (segments are ordered by batch, l)
                    lprev = l2;
                    const int nn = (2 * l2 + 1) * (2 * l2 + 1);
                    const float* __restrict__ D = A.wig + erow * A.nW + is_pick_wig_off(A, l2);
                    const int nj = (nn + 3) >> 2;
#pragma unroll 1
                    for (int j = wave; j < nj; j += NW) {      // image [m * NCO + a][edge]
                        int idx = 4 * j + g;
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
            case 0: epilogue_is<0, NW, SPLIT>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 1: epilogue_is<1, NW, SPLIT>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 2: epilogue_is<2, NW, SPLIT>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 3: epilogue_is<3, NW, SPLIT>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 4: epilogue_is<4, NW, SPLIT>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 5: epilogue_is<5, NW, SPLIT>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 6: epilogue_is<6, NW, SPLIT>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            default: break;
        }
    }
    IS_T(7);                                                   // epilogue
#ifdef HG_PROF
    if (lane == 0) {
        for (int k = 0; k < 12; ++k) atomicAdd(&hg_prof_is_acc[k], prof.t[k]);
        atomicAdd(&hg_prof_is_acc[15], prof.last - t_begin);
    }
#endif
}

#ifdef HG_PROF
extern "C" int hg_prof_is_read(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(hg_prof_is_acc), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(hg_prof_is_acc), z, sizeof(z));
    }
    return 0;
}
#endif

// Compile-time shape of the loaded library, for the host planner to check its own settings against (plan.IS_WAVES / IS_WAVES_LITE / LITE_SRING
// are environment-tunable for A/B builds; a schedule dealt to 8 streams would be misread by a 4-wave kernel -- ADVICE r4).
extern "C" int hg_build_config(int what) {
    switch (what) {
        case 0: return IS_NW;
        case 1: return IS_NW_LITE;
        case 2: return SL_RING;
        default: return -1;
    }
}

extern "C" int hg_tp_is(const float* const* src, const int64_t* src_stride, int nsrc, const float* h2_node, const float* h2_edge,
                        int hidden, const float* wig, int nW, const int32_t* wig_off, const float* weights, const int32_t* seg_table,
                        const int32_t* block_table, const int32_t* phase_table, const int32_t* group_table,
                        const int32_t* item_table, const int32_t* part_table, const int32_t* part_table_host, int nparts,
                        const int32_t* row_table, int lds_bytes,
                        const int64_t* const* src_idx, int rot_mask, const int64_t* edge_perm, const int32_t* run_id, float* out, int64_t out_stride,
                        int64_t rows, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nsrc < 1 || nsrc > 4) return hg_fail(-2, "hg_tp_is: nsrc must be 1..4");
    if (hidden & 15) return hg_fail(-2, "hg_tp_is: (padded) hidden width must be a multiple of 16");
    if (lds_bytes <= 0 || lds_bytes > 160 * 1024) return hg_fail(-2, "hg_tp_is: bad LDS size");
    if (!part_table || !part_table_host || nparts < 1 || nparts > 64) return hg_fail(-2, "hg_tp_is: 1..64 parts");
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
    const int32_t* p0 = part_table_host;                       // single-part launches take the schedule scalars as kernel arguments
    A.nseg = p0[1], A.nphase = p0[3], A.trash_off = p0[4], A.stage_off = p0[5], A.ctr_off = p0[6];
    A.rowtab_off = p0[8], A.rowtab_begin = p0[9], A.rowtab_len = p0[10];
    A.s_split = p0[12] == 1 && hidden == 64;
    A.s_scale = ldexpf(1.f, -(p0[13] + IS_SPLIT_H_EXP));       // S = s_scale (S0 + 2^-11 S1): the weights' 2^sw (part record [13]) and the hidden rows' 2^sh undone
    if (!row_table) return hg_fail(-2, "hg_tp_is: no row table");
    for (int p = 0; p < nparts; ++p) {
        const int32_t* q = part_table_host + p * IS_PART_I32;
        if (q[6] < q[5] || q[5] < q[8] + q[10] || q[8] < q[4] || lds_bytes < 4 * (q[6] + 1) || (q[7] && (part_table_host[11] ? IS_NW_LITE : IS_NW) * q[7] > q[8]))
            return hg_fail(-2, "hg_tp_is: bad LDS layout");
    }
    for (int i = 0; i < 4; ++i) A.idx[i] = (src_idx && i < nsrc) ? src_idx[i] : nullptr;
    A.rot_mask = rot_mask;
    A.eperm = edge_perm;
    A.run_id = run_id;
    if (run_id && nparts != 1) return hg_fail(-2, "hg_tp_is: the fused scatter needs a single-part launch");
    if (rot_mask && !wig) return hg_fail(-2, "hg_tp_is: rotated sources need the Wigner rows");
    static unsigned char lds_attr_done[4][HG_MAX_DEVICES];     // once per device (not a stream operation: illegal during graph capture)
    if (int rc = hg_lds_attr_once(lds_attr_done[0], dev_guard.dev, (const void*)tp_is_kernel<false, false>, 160 * 1024)) return rc;
    if (int rc = hg_lds_attr_once(lds_attr_done[1], dev_guard.dev, (const void*)tp_is_kernel<true, false>, 160 * 1024)) return rc;
    if (int rc = hg_lds_attr_once(lds_attr_done[2], dev_guard.dev, (const void*)tp_is_kernel<false, true>, 160 * 1024)) return rc;
    if (int rc = hg_lds_attr_once(lds_attr_done[3], dev_guard.dev, (const void*)tp_is_kernel<true, true>, 160 * 1024)) return rc;
    const unsigned grid = (unsigned)((rows + 15) / 16);
    const bool lite = p0[11] != 0;                             // part record [11]: the program holds lite_mode items (plan.is_schedule)
#define IS_LAUNCH(SPLITv, LITEv, GRID) \
    hipLaunchKernelGGL((tp_is_kernel<SPLITv, LITEv>), GRID, dim3(LITEv ? 64 * IS_NW_LITE : IS_NT), lds_bytes, (hipStream_t)stream, A, seg_table, block_table, phase_table, \
                       group_table, item_table, weights, part_table, row_table)
    if (nparts == 1) {
        if (lite) IS_LAUNCH(false, true, dim3(grid));
        else IS_LAUNCH(false, false, dim3(grid));
    } else {
        if (lite) IS_LAUNCH(true, true, dim3(grid, (unsigned)nparts));
        else IS_LAUNCH(true, false, dim3(grid, (unsigned)nparts));
    }
#undef IS_LAUNCH
    return hg_check_launch("hg_tp_is");
}

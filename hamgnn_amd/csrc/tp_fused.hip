// tp_fused.hip -- the fused equivariant edge kernel for MI355X (gfx950, CDNA4).  Hand-written HIP, no portability layer.
//
// One launch = one whole reference MessagePackBlock.forward (hamgnn/nn/message_passing.py:191-231) in the edge-aligned
// frame (hamgnn_amd/so3.py): for every output irrep k ("segment") and every (input irrep i, k) super-path row chunk
// ("item", built by hamgnn_amd/plan.py):
//      GEMM1  mid[(l_sh,w), m] = sum_u  (c_p W_p[u,w]) * x'_i[u, src(m)]         v_mfma_f32_16x16x4_f32, A = weights
//      scale  mid *= s_e[(l_sh,w)] * coef_p[m]        s_e = last radial-MLP layer, also an MFMA (K = hidden width)
//      GEMM2  out'_k[w'', m] += sum_rows L'_k[row, w''] * mid[row, m]            A = folded Linear weights
// Edges are the MFMA *columns* (16 per wave), channels are the rows.  Because C/D fragments hold for lane (col=e, g)
// the rows 4g+r, a C register can be fed straight back as the next MFMA's B operand (k = g) with the A operand packed
// in the matching permuted-K order: GEMM1 -> scale -> GEMM2 chain entirely in registers, no cross-lane traffic.
// The per-segment output tile lives in wave-private LDS ([row][m][16 edges], row stride = 16*nco+4 floats so that the
// four row-groups g of a C fragment hit disjoint bank halves); the epilogue optionally applies D^l(R_e)^T (messages
// leave the edge frame before the node scatter) and writes planar rows.
//
// Workgroup = 256 threads = 4 waves (one per SIMD); wave w owns edges [64*block + 16*w, +16).  All operands stream from
// L1/L2: weights are pre-packed in fragment order (one coalesced 256-B load per MFMA A operand), B operands are dword
// loads of the planar rotated feature rows (4 consecutive channels per 16-B segment).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TpArgs {
    const float* src[4];
    int64_t sstride[4];
    const float* h2[2];
    int hidden;
    const float* wig;
    int nW;
    int wig_off[8];
    const float* W;
    const int* segs;
    int nseg;
    const int* items;
    float* out;
    int64_t ostride;
    int64_t rows;
    int tile_floats_wave;
    const float* res[2];         // optional residual rows (same planar layout as `out`), added in the epilogue: out = program(x) + res0 + res1
    int64_t rstride[2];
};

#define SEG_UNROTATE 1
#ifndef HG_TP_WAVES
#define HG_TP_WAVES 2            // min waves per SIMD the register allocator leaves room for (measured: 2 beats 1, 3, 4 - r1 A/B)
#endif

__device__ __forceinline__ const float* pick_src(const TpArgs& A, int i) {
    return i == 0 ? A.src[0] : (i == 1 ? A.src[1] : (i == 2 ? A.src[2] : A.src[3]));
}
__device__ __forceinline__ int64_t pick_stride(const TpArgs& A, int i) {
    return i == 0 ? A.sstride[0] : (i == 1 ? A.sstride[1] : (i == 2 ? A.sstride[2] : A.sstride[3]));
}

// phase profiler (HG_PROF builds only, tests/bench_tp.py --prof): per-wave shader-clock time between probes, summed over waves
#ifdef HG_PROF
__device__ unsigned long long hg_prof_acc[16];
struct Prof { unsigned long long t[12]; unsigned long long last; };
#define HG_PROF_ARG , Prof& prof
#define HG_PROF_PASS , prof
#define HG_T(k)                                                      \
    do {                                                             \
        const unsigned long long t_ = __builtin_readcyclecounter();  \
        prof.t[k] += t_ - prof.last;                                 \
        prof.last = t_;                                              \
    } while (0)
#else
#define HG_PROF_ARG
#define HG_PROF_PASS
#define HG_T(k)
#endif

// ablation hooks (timing experiments only, tests/build_variants.sh): HG_SINK keeps a value alive without using it
#define HG_SINK(v) asm volatile("" ::"v"(v))
#define HG_LDA(p) (*(p))
#ifndef HG_DMA_AUX
#define HG_DMA_AUX 2              // cache policy of the B-operand DMA: nt (streamed once per CU; keeps the shared A lines in L1; r1 A/B: -3 %)
#endif
// tiles and DMA rings are wave-private: the four waves of a workgroup never exchange data, so a wave-local LDS fence
// (LDS ops of one wave retire in order) replaces workgroup barriers and the waves free-run (their DMA waits interleave).
#define HG_WAVE_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define HG_STAGE_FLOATS 2816      // wave-private LDS-DMA ring for B operands: 11 KiB = 11 x 1-KiB (float4) or 44 x 256-B (dword) slots

// LDS-DMA: per-lane global address -> LDS at (wave-uniform base + lane * size); no VGPR round trip, counted by vmcnt
__device__ __forceinline__ void hg_dma16(const float* __restrict__ gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, HG_DMA_AUX);
}
__device__ __forceinline__ void hg_dma4(const float* __restrict__ gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 4, 0, 0);
}

// B-operand span of one (item, source): NC * mulp contiguous floats per edge row -> linear LDS image (see item_body).
// Sources of a two-source item share the ring when both spans fit (slot = source index), else they take turns at offset 0.
#define HG_RING_PIECES 44
struct Span {                      // wave-uniform (SGPR) description of an item's B operand; s0 < 0: none
    int s0, s1, in_off, in_mulp, li, mm;
};
__device__ __forceinline__ int span_nj(const Span& sp) { return ((2 * sp.mm + 1) * (sp.in_mulp >> 2) + 3) >> 2; }
__device__ __forceinline__ bool span_both_fit(const Span& sp) { return sp.s1 >= 0 && 2 * span_nj(sp) * 4 <= HG_RING_PIECES; }
__device__ __forceinline__ int span_slot_floats(const Span& sp, int si) { return span_both_fit(sp) ? si * span_nj(sp) * 256 : 0; }
__device__ __forceinline__ void issue_span(const TpArgs& A, const Span& sp, int si, float* __restrict__ stage, int64_t erow, int g) {
    const int sidx = si ? sp.s1 : sp.s0;
    const int P = (2 * sp.mm + 1) * (sp.in_mulp >> 2), nj = (P + 3) >> 2;
    const float* __restrict__ row = pick_src(A, sidx) + erow * pick_stride(A, sidx) + sp.in_off + (sp.li - sp.mm) * sp.in_mulp;
    float* __restrict__ dst = stage + span_slot_floats(sp, si);
#pragma unroll 1
    for (int j = 0; j < nj; ++j) {
        int p = 4 * j + g;
        p = p < P ? p : P - 1;
        hg_dma16(row + 4 * p, dst + j * 256);
    }
}

template <int MM, int RTM>
__device__ __forceinline__ void item_body(const TpArgs& A, const float* __restrict__ Wb, const int* __restrict__ it,
                                          float* __restrict__ tile, float* __restrict__ stage, int rowstride, int lk, int rto, int mul_k,
                                          int64_t erow, int lane HG_PROF_ARG) {
    constexpr int NC = 2 * MM + 1;
    constexpr int CW = NC > 7 ? (NC + 1) / 2 : NC;             // GEMM2 column chunk (keeps its accumulators <= 28 VGPRs)
    const int typ = it[0], s0 = it[1], s1 = it[2], in_off = it[3], in_mulp = it[4], li = it[5], neg = it[7];
    const int ksteps = it[8], mlp = it[10], x4 = it[17], nk2 = it[18];
    const int g = lane >> 4, el = lane & 15;

    // ---------------------------------------------------------------- GEMM1: mid = A1 fragments x B(rotated features)
    // A operand: one float4 per lane = 4 K-steps (pre-packed, coalesced 1 KiB per wave-load), register double-buffered.
    // B operand: per item and source ONE contiguous span of NC*mulp floats per edge row (components a_lo .. a_lo+NC-1 of the
    // input irrep), staged by LDS-DMA (global_load_lds: no landing VGPRs, all requests of the item in flight at once, ONE
    // vmcnt(0)).  One DMA instruction = 16 rows x 4 consecutive 16-B pieces (lane = row + 16*(piece&3)), which makes the LDS
    // image linear:  float offset(piece p, row e) = 64 p + 4 e.  A fragment read is then `column base + 64*step`: one pointer
    // per column bumped by a constant (the r1 ISA audit of the previous swizzled image showed 241 instructions per 14 MFMAs
    // in this loop -- issue-bound on address arithmetic, not memory-bound).  ds_read_b128 of 16 rows is conflict-free, the
    // dword reads of x1 mode are 2-way.
    const int nsrc = s1 >= 0 ? 2 : 1;
    const int ngrp = (ksteps + 3) >> 2;
    const int P1 = in_mulp >> 2;                               // float4 pieces per column
    const int P = NC * P1;                                     // pieces per row span (planner guarantees P <= 40)
    const int nj = (P + 3) >> 2;                               // DMA instructions per source
    const int a_lo = li - MM;
    const f32x4* __restrict__ aw = reinterpret_cast<const f32x4*>(Wb + it[11]) + lane;      // [src][G][rt][lane]
    const int cdir = neg ? -P1 : P1;                           // column c -> span piece base (neg ? NC-1-c : c) * P1
    const int c0p = neg ? (NC - 1) * P1 : 0;

    HG_T(0);                                                    // dispatch: record loads, switch, prologue
    const Span me = Span{s0, s1, in_off, in_mulp, li, MM};
    int nissued = 0;                                            // sources of this item whose span DMAs are out
    {   // this item's spans go out first: their (HBM-class) latency runs under the radial-scale phase below
        const int nfirst = nsrc == 0 ? 0 : (span_both_fit(me) ? nsrc : 1);
        if (nissued < nfirst) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // previous fragment reads retired
        for (; nissued < nfirst; ++nissued) issue_span(A, me, nissued, stage, erow, g);
    }

    // ---------------------------------------------------------------- radial scale s_e = W3^T h2  (MFMA, K = hidden, permuted K)
    // Runs BEFORE GEMM1 (it does not depend on it): `mid` is not live yet, so every W3 / h fragment of the item (up to 4 x RTM + 4
    // float4) is requested at once -- ONE exposed L2 latency, overlapped with the span DMAs, instead of one per 16-wide K group
    // (r1b ablation: weight-load latency cost 2.3 of 9.7 ms, most of it in this phase whose MFMA time per fragment is only 128 clk).
    f32x4 S[RTM];
    if (typ == 0) {
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) S[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ hrow = (mlp ? A.h2[1] : A.h2[0]) + erow * A.hidden + 4 * g;
        const f32x4* __restrict__ w3 = reinterpret_cast<const f32x4*>(Wb + it[12]) + lane;
        const int hgrp = A.hidden >> 4;
#pragma unroll 1
        for (int G0 = 0; G0 < hgrp; G0 += 4) {
            f32x4 hb[4], wv[4][RTM];
#pragma unroll
            for (int G = 0; G < 4; ++G)
                if (G0 + G < hgrp) {
                    hb[G] = HG_LDA(reinterpret_cast<const f32x4*>(hrow + 16 * (G0 + G)));
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt) wv[G][rt] = HG_LDA(w3 + ((G0 + G) * RTM + rt) * 64);
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

    HG_T(1);                                                    // span issue + radial-scale phase
    f32x4 mid[RTM][NC];
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
        for (int c = 0; c < NC; ++c) mid[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 av_n[RTM];
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt) av_n[rt] = HG_LDA(aw + rt * 64);
#pragma unroll 1
    for (int si = 0; si < nsrc; ++si) {
        if (si >= nissued) {                                    // second source of a span pair too wide to share the ring
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_span(A, me, si, stage, erow, g);
            ++nissued;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HG_T(2);                                                // waiting for the spans (and the first A fragments)
        const float* __restrict__ sbase = stage + span_slot_floats(me, si);
        const int abase = si * ngrp;
        if (NC <= 3 && x4) {                                   // permuted K: fragment (c, G) = piece cbase + 4G + g of row el
            const float* __restrict__ fb = sbase + (c0p + g) * 64 + el * 4;
#pragma unroll 1
            for (int G = 0; G < ngrp; ++G) {
                f32x4 av[RTM], bv[NC];
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av[rt] = av_n[rt];
                if (abase + G + 1 < nsrc * ngrp) {
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt) av_n[rt] = HG_LDA(aw + ((abase + G + 1) * RTM + rt) * 64);
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) bv[c] = *reinterpret_cast<const f32x4*>(fb + (c * cdir + 4 * G) * 64);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            mid[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][q], bv[c][q], mid[rt][c], 0, 0, 0);
            }
        } else {                                               // natural K: element (c, 4 sl + g) = piece cbase + sl, component g
            const float* __restrict__ fb = sbase + c0p * 64 + el * 4 + g;
#pragma unroll 1
            for (int G = 0; G < ngrp; ++G) {
                f32x4 av[RTM];
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av[rt] = av_n[rt];
                if (abase + G + 1 < nsrc * ngrp) {
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt) av_n[rt] = HG_LDA(aw + ((abase + G + 1) * RTM + rt) * 64);
                }
                const int nq = ksteps - 4 * G;                 // K-steps in this group (>= 4 except in the tail group)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < nq) {
                        float b[NC];
#pragma unroll
                        for (int c = 0; c < NC; ++c) b[c] = fb[(c * cdir + 4 * G + q) * 64];
#pragma unroll
                        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                mid[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][q], b[c], mid[rt][c], 0, 0, 0);
                    }
                }
            }
        }
        HG_T(3);                                                // GEMM1 of this source
    }

    if (typ == 0) {
        const f32x4* __restrict__ a2 = reinterpret_cast<const f32x4*>(Wb + it[14]) + lane;
        const f32x4* __restrict__ cf = reinterpret_cast<const f32x4*>(Wb + it[13]) + g;     // [rt][c][g] float4
        {
        f32x4 a2_n[RTM];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = HG_LDA(a2 + rt * 64);
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int c = 0; c < NC; ++c) mid[rt][c] = mid[rt][c] * S[rt] * HG_LDA(cf + (rt * NC + c) * 4);

        HG_T(4);                                                // mid *= S * cf (+ a2 / cf loads)
        // ------------------------------------------------------------ GEMM2: tile[w'', m] += L' fragments x mid
        const int rto_run = rto;
#pragma unroll 1
        for (int rtp = 0; rtp < rto_run; ++rtp) {
            f32x4 av[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = a2_n[rt];
            if (rtp + 1 < rto) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = HG_LDA(a2 + ((rtp + 1) * RTM + rt) * 64);
            }
            // rows beyond mul_k (fragment padding) are redirected to a trash row behind the tile: no divergent branches
            float* __restrict__ trow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * rtp + 4 * g + r;
                trow[r] = tile + (rr < mul_k ? rr : mul_k) * rowstride + (lk - MM) * 16 + el;
            }
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += CW) {
                // the tile values are the accumulator init (C operand): no separate add
                f32x4 acc[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c0 + c < NC) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[c][r] = trow[r][(c0 + c) * 16];
                    }
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * rt + r < nk2) {                // trailing K-steps hold only padding rows (planner row order): not issued
#pragma unroll
                            for (int c = 0; c < CW; ++c)
                                if (c0 + c < NC)
                                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][r], mid[rt][c0 + c][r], acc[c], 0, 0, 0);
                        }
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c0 + c < NC) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) trow[r][(c0 + c) * 16] = acc[c][r];
                    }
            }
        }
        }
        HG_T(5);                                                // GEMM2 + tile write-back
    } else {
        // plain o3.Linear path: rows are output channels; add straight into the tile
        // (typ 2, lite_mode paths: each column first takes its aligned-frame CG coefficient, message_passing.py:197-215)
        const int row0 = it[16];
        if (typ == 2) {
            const float* __restrict__ cfc = Wb + it[13];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float cv = cfc[c];
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) mid[rt][c] = mid[rt][c] * cv;
            }
        }
        float* __restrict__ t0[RTM][4];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = row0 + 16 * rt + 4 * g + r;
                t0[rt][r] = tile + (rr < mul_k ? rr : mul_k) * rowstride + (lk - MM) * 16 + el;
#pragma unroll
                for (int c = 0; c < NC; ++c) mid[rt][c][r] += t0[rt][r][c * 16];
            }
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) t0[rt][r][c * 16] = mid[rt][c][r];
        HG_T(6);                                                // linear-item write-back
    }
}

// lite_mode segment post-op (message_passing.py:209-215: combine_messages = LinearScaleWithWeights on the summed branches):
//   tile[v, m] <- sum_w'' Lc[w'', v] * s_e[w''] * tile[w'', m],   s_e = W3^T h2 by MFMA, in place per column chunk.
__device__ __forceinline__ void post_item(const TpArgs& A, const float* __restrict__ Wb, const int* __restrict__ it,
                                          float* __restrict__ tile, int rowstride, int nco, int rto, int mul_k, int64_t erow, int lane) {
    const int g = lane >> 4, el = lane & 15;
    const float* __restrict__ hrow = A.h2[0] + erow * A.hidden + 4 * g;
    const f32x4* __restrict__ w3 = reinterpret_cast<const f32x4*>(Wb + it[12]) + lane;          // [G][rto][lane]
    const f32x4* __restrict__ a2 = reinterpret_cast<const f32x4*>(Wb + it[14]) + lane;          // [rtp][rt][lane]
    f32x4 S[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) S[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int hgrp = A.hidden >> 4;
#pragma unroll 1
    for (int G = 0; G < hgrp; ++G) {
        const f32x4 hb = *reinterpret_cast<const f32x4*>(hrow + 16 * G);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if (rt < rto) {
                const f32x4 wv = w3[(G * rto + rt) * 64];
#pragma unroll
                for (int q = 0; q < 4; ++q) S[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[q], hb[q], S[rt], 0, 0, 0);
            }
        }
    }
    int rowoff[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 16 * rt + 4 * g + r;
            rowoff[rt][r] = (rr < mul_k ? rr : mul_k) * rowstride + el;
        }
#pragma unroll 1
    for (int c0 = 0; c0 < nco; c0 += 4) {
        f32x4 md[4][4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                md[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (rt < rto && c0 + c < nco) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) md[rt][c][r] = (16 * rt + 4 * g + r < mul_k) ? tile[rowoff[rt][r] + (c0 + c) * 16] * S[rt][r] : 0.f;
                }
            }
        HG_WAVE_FENCE();
#pragma unroll 1
        for (int rtp = 0; rtp < rto; ++rtp) {
            f32x4 acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < rto) {
                    const f32x4 av = a2[(rtp * rto + rt) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], md[rt][c][r], acc[c], 0, 0, 0);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c0 + c < nco) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = 16 * rtp + 4 * g + r;
                        tile[(rr < mul_k ? rr : mul_k) * rowstride + el + (c0 + c) * 16] = acc[c][r];
                    }
                }
        }
    }
}

template <int LK>
__device__ __forceinline__ void epilogue(const TpArgs& A, const float* __restrict__ tile, float* __restrict__ stage, int rowstride, int mul_k,
                                         int out_off, int out_mulp, int flags, int64_t e, int64_t erow, bool valid, int lane) {
    constexpr int NCO = 2 * LK + 1;
    const int g = lane >> 4, el = lane & 15;
    const float* __restrict__ tl = tile + el;
    float* __restrict__ ob = A.out + e * A.ostride + out_off;
    const int wend = mul_k + (flags >> 8);                     // + channel-padding slots of the planar block (last chunk only)
    if (flags & SEG_UNROTATE) {
        // out[w, a] = sum_m D^l(R_e)[m, a] tile[w, m].  The 16 Wigner blocks of the wave are pulled into the (idle) B-operand
        // ring by LDS-DMA, image [m * NCO + a][edge]: ONE exposed memory latency per segment (the r1 ablations showed per-column
        // global loads costing 0.6-0.75 of 10 ms: ~50 exposed L2/HBM latencies per wave), conflict-free broadcast reads after.
        const float* __restrict__ D = A.wig + erow * A.nW + A.wig_off[LK];
        constexpr int NJ = (NCO * NCO + 3) / 4;
#pragma unroll 1
        for (int j = 0; j < NJ; ++j) {
            int idx = 4 * j + g;
            idx = idx < NCO * NCO ? idx : NCO * NCO - 1;
            hg_dma4(D + idx, stage + j * 64);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float* __restrict__ dl = stage + el;
#pragma unroll 1
        for (int a = 0; a < NCO; ++a) {
            float dc[NCO];
#pragma unroll
            for (int m = 0; m < NCO; ++m) dc[m] = dl[(m * NCO + a) * 16];
#pragma unroll 1
            for (int w = g; w < wend; w += 4) {                // rows mul_k .. wend-1 are the block's channel padding: written as zeros
                const float* __restrict__ tw = tl + (w < mul_k ? w : mul_k) * rowstride;
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < NCO; ++m) acc = fmaf(dc[m], tw[m * 16], acc);
                if (valid) ob[a * out_mulp + w] = w < mul_k ? acc : 0.f;
            }
        }
    } else {
        // Planar store (+ optional residual rows: the x + Lin2(Gate(Lin1 x)) [+ skip] adds of ResidualBlock / ConvBlockE3,
        // hamgnn/nn/interaction_blocks.py:352-357, convolution.py:158, ride on the producing launch).  A lane owns FOUR consecutive
        // channels of one (row, component): the four lane groups of a row write 64 contiguous bytes per store instruction (whole
        // sectors; the r1 epilogue wrote 4 scattered dwords per row and instruction and ran the E-row Linears at 1.3 TB/s), and the
        // residual rows are read the same way.
        const float* __restrict__ r0 = A.res[0] ? A.res[0] + erow * A.rstride[0] + out_off : nullptr;
        const float* __restrict__ r1 = A.res[1] ? A.res[1] + erow * A.rstride[1] + out_off : nullptr;
#pragma unroll 1
        for (int w0 = 4 * g; w0 < wend; w0 += 16) {            // wend and out_mulp are multiples of 4 except for the last chunk of a split irrep
            if (w0 + 4 <= wend) {
#pragma unroll
                for (int a = 0; a < NCO; ++a) {
                    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (r0) v = *reinterpret_cast<const f32x4*>(r0 + a * out_mulp + w0);
                    if (r1) v += *reinterpret_cast<const f32x4*>(r1 + a * out_mulp + w0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (w0 + j < mul_k) ? v[j] + tl[(w0 + j) * rowstride + a * 16] : 0.f;
                    if (valid) *reinterpret_cast<f32x4*>(ob + a * out_mulp + w0) = v;
                }
            } else {                                           // ragged tail (segment chunks of a split irrep that are not multiples of 4)
#pragma unroll 1
                for (int w = w0; w < wend; ++w)
#pragma unroll
                    for (int a = 0; a < NCO; ++a) {
                        float v = 0.f;
                        if (w < mul_k) v = tl[w * rowstride + a * 16] + (r0 ? r0[a * out_mulp + w] : 0.f) + (r1 ? r1[a * out_mulp + w] : 0.f);
                        if (valid) ob[a * out_mulp + w] = v;
                    }
            }
        }
    }
}

#define HG_CASE(MMv, RTMv) \
    case (MMv * 8 + RTMv): item_body<MMv, RTMv>(A, g_W, it, tile, stage, rowstride, lk, rto, mul_k, erow, lane HG_PROF_PASS); break;

template <bool HAS_POST>
__global__ __launch_bounds__(256, HG_TP_WAVES) void tp_fused_kernel(const TpArgs A, const int* __restrict__ g_segs, const int* __restrict__ g_items,
                                                                   const float* __restrict__ g_W) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 64 + wave * 16 + (lane & 15);
    const bool valid = e < A.rows;
    const int64_t erow = valid ? e : A.rows - 1;
    float* tile = lds + wave * A.tile_floats_wave;
    float* stage = tile + (A.tile_floats_wave - HG_STAGE_FLOATS);          // B-operand DMA ring behind the segment tile
#ifdef HG_PROF
    Prof prof;
    for (int k = 0; k < 12; ++k) prof.t[k] = 0;
    prof.last = __builtin_readcyclecounter();
    const unsigned long long t_begin = prof.last;
#endif

    // segments are independent (each owns its tile and items): launches with few row tiles put one segment per workgroup (grid.y) so
    // that a 2-atom crystal's Linear is not one workgroup walking every segment serially
    const int sg_begin = gridDim.y > 1 ? blockIdx.y : 0, sg_end = gridDim.y > 1 ? blockIdx.y + 1 : A.nseg;
    for (int sg = sg_begin; sg < sg_end; ++sg) {
        const int* __restrict__ S = g_segs + sg * 8;
        const int lk = S[0], mul_k = S[1], rto = S[2], out_off = S[3], out_mulp = S[4], ib = S[5], ie = S[6], flags = S[7];
        const int nco = 2 * lk + 1;
        const int rowstride = nco * 16 + 4;
        const int tfl = (mul_k + 1) * rowstride;           // + trash row for padded fragment rows
        for (int i = lane; i < tfl; i += 64) tile[i] = 0.f;
        HG_WAVE_FENCE();
        HG_T(7);                                               // segment set-up: tile zeroing
        for (int ii = ib; ii < ie; ++ii) {
            const int* __restrict__ it = g_items + ii * 20;
            const int mm = it[6], rtm = it[9];
            // the last item of a segment stages nothing ahead: the epilogue borrows the ring for the Wigner blocks
            if (HAS_POST && it[0] == 3) {
                // segment post-op (lite_mode programs only: separate instantiation)
                HG_WAVE_FENCE();
                post_item(A, g_W, it, tile, rowstride, nco, rto, mul_k, erow, lane);
                continue;
            }
            switch (mm * 8 + rtm) {
                HG_CASE(0, 1) HG_CASE(0, 2) HG_CASE(0, 3) HG_CASE(0, 4)
                HG_CASE(1, 1) HG_CASE(1, 2) HG_CASE(1, 3) HG_CASE(1, 4)
                HG_CASE(2, 1) HG_CASE(2, 2) HG_CASE(2, 3)
                HG_CASE(3, 1) HG_CASE(3, 2)
                HG_CASE(4, 2)                                              // row-tile table of plan.py:rtm_max (4,4,3,2,2,1,1)
                HG_CASE(4, 1)
                HG_CASE(5, 1)
                HG_CASE(6, 1)
                default: break;
            }
        }
        HG_WAVE_FENCE();
        // opaque per segment: the row-derived pointers of the epilogue (output, residual rows, Wigner blocks) are rebuilt here instead of
        // being hoisted out of the segment loop and carried -- spilled, in the lite_mode instantiation -- through all items
        int64_t e_o = e, erow_o = erow;
        asm volatile("" : "+v"(e_o), "+v"(erow_o));
#define e e_o
#define erow erow_o
        switch (lk) {
            case 0: epilogue<0>(A, tile, stage, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 1: epilogue<1>(A, tile, stage, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 2: epilogue<2>(A, tile, stage, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 3: epilogue<3>(A, tile, stage, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 4: epilogue<4>(A, tile, stage, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 5: epilogue<5>(A, tile, stage, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 6: epilogue<6>(A, tile, stage, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            default: break;
        }
#undef e
#undef erow
        HG_WAVE_FENCE();
        HG_T(8);                                               // epilogue
    }
#ifdef HG_PROF
    if (lane == 0) {
        for (int k = 0; k < 9; ++k) atomicAdd(&hg_prof_acc[k], prof.t[k]);
        atomicAdd(&hg_prof_acc[15], prof.last - t_begin);
    }
#endif
}

#ifdef HG_PROF
extern "C" int hg_prof_read(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(hg_prof_acc), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(hg_prof_acc), z, sizeof(z));
    }
    return 0;
}
#endif

extern "C" int hg_tp_fused(const float* const* src, const int64_t* src_stride, int nsrc, const float* h2_node,
                           const float* h2_edge, int hidden, const float* wig, int nW, const int32_t* wig_off,
                           const float* weights, const int32_t* seg_table, int nseg, const int32_t* item_table, float* out,
                           int64_t out_stride, int64_t rows, int lds_bytes, int program_flags, const float* res0, int64_t res0_stride,
                           const float* res1, int64_t res1_stride, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nsrc < 1 || nsrc > 4) return hg_fail(-2, "hg_tp_fused: nsrc must be 1..4");
    if (hidden & 15) return hg_fail(-2, "hg_tp_fused: (padded) hidden width must be a multiple of 16");
    if (lds_bytes <= 0 || lds_bytes > 160 * 1024) return hg_fail(-2, "hg_tp_fused: bad LDS size");
    TpArgs A;
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
    A.W = weights;
    A.segs = seg_table;
    A.nseg = nseg;
    A.items = item_table;
    A.out = out;
    A.ostride = out_stride;
    A.rows = rows;
    A.tile_floats_wave = lds_bytes / 16;           // 4 waves x 4 bytes
    if (res1 && !res0) return hg_fail(-2, "hg_tp_fused: res1 without res0");
    A.res[0] = res0, A.res[1] = res1, A.rstride[0] = res0_stride, A.rstride[1] = res1_stride;
    static unsigned char lds_attr_done[2][HG_MAX_DEVICES];     // once per device (not a stream operation: illegal during graph capture)
    if (int rc = hg_lds_attr_once(lds_attr_done[0], dev_guard.dev, (const void*)tp_fused_kernel<true>, 160 * 1024)) return rc;
    if (int rc = hg_lds_attr_once(lds_attr_done[1], dev_guard.dev, (const void*)tp_fused_kernel<false>, 160 * 1024)) return rc;
    const unsigned grid = (unsigned)((rows + 63) / 64);
    const unsigned gy = ((int64_t)grid * nseg <= 1024 && nseg > 1) ? (unsigned)nseg : 1u;      // few row tiles: one segment per workgroup
    if (program_flags & 1)
        hipLaunchKernelGGL(tp_fused_kernel<true>, dim3(grid, gy), dim3(256), lds_bytes, (hipStream_t)stream, A, seg_table, item_table, weights);
    else
        hipLaunchKernelGGL(tp_fused_kernel<false>, dim3(grid, gy), dim3(256), lds_bytes, (hipStream_t)stream, A, seg_table, item_table, weights);
    return hg_check_launch("hg_tp_fused");
}

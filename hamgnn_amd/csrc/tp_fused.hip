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

template <int MM, int RTM>
__device__ __forceinline__ void item_body(const TpArgs& A, const int* __restrict__ it, float* __restrict__ tile,
                                          int rowstride, int lk, int rto, int mul_k, int64_t erow, int lane) {
    constexpr int NC = 2 * MM + 1;
    constexpr int CW = NC > 7 ? (NC + 1) / 2 : NC;             // GEMM2 column chunk (keeps its accumulators <= 28 VGPRs)
    const int typ = it[0], s0 = it[1], s1 = it[2], in_off = it[3], in_mulp = it[4], li = it[5], neg = it[7];
    const int ksteps = it[8], mlp = it[10], x4 = it[17];
    const int g = lane >> 4, el = lane & 15;

    f32x4 mid[RTM][NC];
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
        for (int c = 0; c < NC; ++c) mid[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---------------------------------------------------------------- GEMM1: mid = A1 fragments x B(rotated features)
    // A operand: one float4 per lane = 4 K-steps (pre-packed, coalesced 1 KiB per wave-load).
    // B operand: x4 mode (mulp % 16 == 0, NC <= 3): one float4 per (column, 4 K-steps), permuted K order u = 16G + 4g + q;
    //            x1 mode: one dword per (column, K-step), u = 4s + g.
    // Register double buffering: the fragments of group t+1 are requested before the MFMAs of group t are issued, so the
    // L2/HBM latency of the operand stream hides behind 4*RTM*NC MFMAs even at 2 waves per SIMD.
    const int step = neg ? -in_mulp : in_mulp;                 // column c <-> m = c - MM, input component a = li +/- m
    const int a0 = neg ? li + MM : li - MM;
    const int ngrp = (ksteps + 3) >> 2;
    const int nsrc = s1 >= 0 ? 2 : 1;
    const int ntot = nsrc * ngrp;
    const f32x4* __restrict__ aw = reinterpret_cast<const f32x4*>(A.W + it[11]) + lane;      // [src][G][rt][lane]
    const float* __restrict__ xsrc0 = pick_src(A, s0) + erow * pick_stride(A, s0) + in_off + a0 * in_mulp;
    const float* __restrict__ xsrc1 = nsrc == 2 ? pick_src(A, s1) + erow * pick_stride(A, s1) + in_off + a0 * in_mulp : xsrc0;

    // early requests for the later phases (their latency hides behind GEMM1)
    const float* __restrict__ hrow = (mlp ? A.h2[1] : A.h2[0]) + erow * A.hidden + 4 * g;
    const f32x4* __restrict__ w3 = reinterpret_cast<const f32x4*>(A.W + it[12]) + lane;
    f32x4 hb_n = (f32x4){0.f, 0.f, 0.f, 0.f}, w3_n[RTM];
    if (typ == 0) {
        hb_n = *reinterpret_cast<const f32x4*>(hrow);
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) w3_n[rt] = w3[rt * 64];
    }

    if (NC <= 3 && x4) {                                       // planner sets x4 only for NC <= 3 (keeps bv[] at 12 VGPRs)
        f32x4 av_n[RTM], bv_n[NC];
        {
            const float* __restrict__ x0 = xsrc0 + 4 * g;
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[rt * 64];
#pragma unroll
            for (int c = 0; c < NC; ++c) bv_n[c] = *reinterpret_cast<const f32x4*>(x0 + c * step);
        }
#pragma unroll 1
        for (int t = 0; t < ntot; ++t) {
            f32x4 av[RTM], bv[NC];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = av_n[rt];
#pragma unroll
            for (int c = 0; c < NC; ++c) bv[c] = bv_n[c];
            if (t + 1 < ntot) {
                const int tn = t + 1;
                const int G = tn >= ngrp ? tn - ngrp : tn;
                const float* __restrict__ x0 = (tn >= ngrp ? xsrc1 : xsrc0) + 4 * g + 16 * G;
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[(tn * RTM + rt) * 64];
#pragma unroll
                for (int c = 0; c < NC; ++c) bv_n[c] = *reinterpret_cast<const f32x4*>(x0 + c * step);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        mid[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][q], bv[c][q], mid[rt][c], 0, 0, 0);
        }
    } else {
        // x1: K-steps s = 0 .. nsrc*ksteps-1 flattened; A float4 covers steps 4G..4G+3 of one source
        const int nsteps = nsrc * ksteps;
        float b_n[NC];
        f32x4 av[RTM];
        {
            const float* __restrict__ x0 = xsrc0 + g;
#pragma unroll
            for (int c = 0; c < NC; ++c) b_n[c] = x0[c * step];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = aw[rt * 64];
        }
        int sl = 0, srcsel = 0;                                // local step within the source, source index
#pragma unroll 1
        for (int t = 0; t < nsteps; ++t) {
            float b[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) b[c] = b_n[c];
            const int q = sl & 3;
            f32x4 avc[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) avc[rt] = av[rt];
            // advance
            int sl_n = sl + 1, src_n = srcsel;
            if (sl_n == ksteps) { sl_n = 0; src_n = srcsel + 1; }
            if (t + 1 < nsteps) {
                const float* __restrict__ x0 = (src_n ? xsrc1 : xsrc0) + g + 4 * sl_n;
#pragma unroll
                for (int c = 0; c < NC; ++c) b_n[c] = x0[c * step];
                if ((sl_n & 3) == 0) {
                    const int Gn = src_n * ngrp + (sl_n >> 2);
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt) av[rt] = aw[(Gn * RTM + rt) * 64];
                }
            }
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) {
                const float a = q == 0 ? avc[rt][0] : (q == 1 ? avc[rt][1] : (q == 2 ? avc[rt][2] : avc[rt][3]));
#pragma unroll
                for (int c = 0; c < NC; ++c) mid[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[c], mid[rt][c], 0, 0, 0);
            }
            sl = sl_n; srcsel = src_n;
        }
    }

    float* __restrict__ tp = tile + (4 * g) * rowstride + (lk - MM) * 16 + el;
    if (typ == 0) {
        // ------------------------------------------------------------ radial scale s_e = W3^T h2  (MFMA, K = hidden, permuted K)
        f32x4 S[RTM];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) S[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int hgrp = A.hidden >> 4;
        const f32x4* __restrict__ a2 = reinterpret_cast<const f32x4*>(A.W + it[14]) + lane;
        f32x4 a2_n[RTM];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[rt * 64];                               // GEMM2 operands of rtp = 0, early
#pragma unroll 1
        for (int G = 0; G < hgrp; ++G) {
            const f32x4 hb = hb_n;
            f32x4 wv[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) wv[rt] = w3_n[rt];
            if (G + 1 < hgrp) {
                hb_n = *reinterpret_cast<const f32x4*>(hrow + 16 * (G + 1));
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) w3_n[rt] = w3[((G + 1) * RTM + rt) * 64];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) S[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[rt][q], hb[q], S[rt], 0, 0, 0);
        }
        const f32x4* __restrict__ cf = reinterpret_cast<const f32x4*>(A.W + it[13]) + g;     // [rt][c][g] float4
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int c = 0; c < NC; ++c) mid[rt][c] = mid[rt][c] * S[rt] * cf[(rt * NC + c) * 4];

        // ------------------------------------------------------------ GEMM2: tile[w'', m] += L' fragments x mid
#pragma unroll 1
        for (int rtp = 0; rtp < rto; ++rtp) {
            f32x4 av[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = a2_n[rt];
            if (rtp + 1 < rto) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[((rtp + 1) * RTM + rt) * 64];
            }
            float* __restrict__ t = tp + (16 * rtp) * rowstride;
            const int rbase = 16 * rtp + 4 * g;
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += CW) {
                f32x4 acc[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < CW; ++c)
                            if (c0 + c < NC)
                                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][r], mid[rt][c0 + c][r], acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c0 + c < NC) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (rbase + r < mul_k) t[r * rowstride + (c0 + c) * 16] += acc[c][r];
                    }
            }
        }
    } else {
        // plain o3.Linear path: rows are output channels; add straight into the tile
        const int row0 = it[16];
        float* __restrict__ t0 = tp + row0 * rowstride;
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + 16 * rt + 4 * g + r < mul_k) t0[(16 * rt + r) * rowstride + c * 16] += mid[rt][c][r];
    }
}

template <int LK>
__device__ __forceinline__ void epilogue(const TpArgs& A, const float* __restrict__ tile, int rowstride, int mul_k,
                                         int out_off, int out_mulp, int flags, int64_t e, int64_t erow, bool valid, int lane) {
    constexpr int NCO = 2 * LK + 1;
    const int g = lane >> 4, el = lane & 15;
    const float* __restrict__ D = A.wig ? A.wig + erow * A.nW + A.wig_off[LK] : nullptr;
#pragma unroll 1
    for (int w = g; w < mul_k; w += 4) {
        float t[NCO];
#pragma unroll
        for (int m = 0; m < NCO; ++m) t[m] = tile[w * rowstride + m * 16 + el];
        float* __restrict__ o = A.out + e * A.ostride + out_off + w;
        if (flags & SEG_UNROTATE) {
#pragma unroll 1
            for (int a = 0; a < NCO; ++a) {
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < NCO; ++m) acc = fmaf(D[m * NCO + a], t[m], acc);
                if (valid) o[a * out_mulp] = acc;
            }
        } else {
#pragma unroll
            for (int a = 0; a < NCO; ++a)
                if (valid) o[a * out_mulp] = t[a];
        }
    }
}

#define HG_CASE(MMv, RTMv) \
    case (MMv * 8 + RTMv): item_body<MMv, RTMv>(A, it, tile, rowstride, lk, rto, mul_k, erow, lane); break;

__global__ __launch_bounds__(256, HG_TP_WAVES) void tp_fused_kernel(const TpArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 64 + wave * 16 + (lane & 15);
    const bool valid = e < A.rows;
    const int64_t erow = valid ? e : A.rows - 1;
    float* tile = lds + wave * A.tile_floats_wave;

    for (int sg = 0; sg < A.nseg; ++sg) {
        const int* __restrict__ S = A.segs + sg * 8;
        const int lk = S[0], mul_k = S[1], rto = S[2], out_off = S[3], out_mulp = S[4], ib = S[5], ie = S[6], flags = S[7];
        const int nco = 2 * lk + 1;
        const int rowstride = nco * 16 + 4;
        const int tfl = mul_k * rowstride;
        for (int i = lane; i < tfl; i += 64) tile[i] = 0.f;
        __syncthreads();
        for (int ii = ib; ii < ie; ++ii) {
            const int* __restrict__ it = A.items + ii * 20;
            const int mm = it[6], rtm = it[9];
            switch (mm * 8 + rtm) {
                HG_CASE(0, 1) HG_CASE(0, 2) HG_CASE(0, 3) HG_CASE(0, 4)
                HG_CASE(1, 1) HG_CASE(1, 2) HG_CASE(1, 3) HG_CASE(1, 4)
                HG_CASE(2, 1) HG_CASE(2, 2) HG_CASE(2, 3)
                HG_CASE(3, 1) HG_CASE(3, 2)
                HG_CASE(4, 1)
                HG_CASE(5, 1)
                HG_CASE(6, 1)
                default: break;
            }
        }
        __syncthreads();
        switch (lk) {
            case 0: epilogue<0>(A, tile, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 1: epilogue<1>(A, tile, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 2: epilogue<2>(A, tile, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 3: epilogue<3>(A, tile, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 4: epilogue<4>(A, tile, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 5: epilogue<5>(A, tile, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            case 6: epilogue<6>(A, tile, rowstride, mul_k, out_off, out_mulp, flags, e, erow, valid, lane); break;
            default: break;
        }
        __syncthreads();
    }
}

extern "C" int hg_tp_fused(const float* const* src, const int64_t* src_stride, int nsrc, const float* h2_node,
                           const float* h2_edge, int hidden, const float* wig, int nW, const int32_t* wig_off,
                           const float* weights, const int32_t* seg_table, int nseg, const int32_t* item_table, float* out,
                           int64_t out_stride, int64_t rows, int lds_bytes, void* stream) {
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
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)tp_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return hg_fail(-3, hipGetErrorString(e));
    }
    const unsigned grid = (unsigned)((rows + 63) / 64);
    hipLaunchKernelGGL(tp_fused_kernel, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, A);
    return hg_check_launch("hg_tp_fused");
}

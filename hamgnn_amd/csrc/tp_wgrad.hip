// tp_wgrad.hip -- fused WEIGHT gradients of the weighted tensor-product branches of a MessagePackBlock (the backward of
// /root/reference/hamgnn/nn/message_passing.py:191-231 with respect to tensor_product.weight, linear_scaler.linear_out.weight,
// linear_out.weight and -- through gs -- the radial weight generators).  Hand-written HIP for gfx950 (CDNA4), fp32 MFMA.
//
// Weight-stationary where the forward (csrc/tp_is.hip) is input-stationary: a workgroup owns one UNIT = up to four 16-row tiles of one
// super-path (input irrep i -> output irrep k; rows = (e3nn path, mid channel)) of plan.WgFused and a range of 16-edge tiles; its
// accumulators never leave the registers.  Per edge tile, with EDGES as the MFMA M index (A operand = staged rows, B operand = weights):
//     mid^T[e, row] (per column c) = x[e, :, comp(c)] . W[row, :]        B^T[e, row] = g[e, :, col(c)] . L[row, :]        s^T[e, row] = h[e, :] . W3[:, row]
// The C fragments hold, for lane (row, g), the edges 4 g + r: exactly the B-operand layout of a K = 16-edges MFMA, so
//     g_W[u, row] += sum_e x[e, u, comp(c)] * (s cf B)[e, row, c]          g_L[w, row] += sum_e g[e, w, col(c)] * (s cf mid)[e, row, c]
// are issued straight from those registers (A operand = the same staged rows read along the edge axis; no transposition, nothing
// materialised).  gs[e, ch(row)] = sum_c cf mid B is written per edge (last radial layer / hidden-layer gradients: two library GEMMs).
// Staging: the next tile's rows travel through registers (float4 loads issued before the second MFMA phase, written to the other LDS
// buffer after it): one barrier per iteration.  LDS rows are [x source 0 | x source 1 | g | h] with stride == 4 (mod 64) floats: the dword
// reads of (16 edges x 4 K-slots) and of (4 edges x 16 channels) are both conflict-free.
// Partial sums of the splits (blockIdx.y) and of the edge-tile copies inside a workgroup go to separate accumulator blocks; the host adds
// them in a fixed order (bit-reproducible, no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WG_UNIT_I32 32
#define WG_NT 256
// float4 pieces a thread holds in flight while the next tile is staged, by column count: plan.WG_PIECES_OF_NC
#define WG_PIECES_OF(NC) ((NC) <= 9 ? 10 : ((NC) == 11 ? 6 : 5))
// accumulator fragments per source (tiles of 16 channels) by column count: plan.WG_MAXT_OF_NC
#define WG_MAXT_OF(NC) ((NC) <= 7 ? 4 : ((NC) == 9 ? 2 : 1))

struct WgArgs {
    const float* src[4];
    int64_t sstride[4];
    const float* g;
    int64_t gstride;
    const float* h[2];
    int64_t hstride;
    float* gs[2];
    int64_t gsstride[2];
    float* acc;
    int64_t acc_split;           // floats per split
    int64_t rows;                // edges of this launch
    int hidden;
};

__device__ __forceinline__ const float* wg_pick_src(const WgArgs& A, int i) {
    return i == 0 ? A.src[0] : (i == 1 ? A.src[1] : (i == 2 ? A.src[2] : A.src[3]));
}
__device__ __forceinline__ int64_t wg_pick_stride(const WgArgs& A, int i) {
    return i == 0 ? A.sstride[0] : (i == 1 ? A.sstride[1] : (i == 2 ? A.sstride[2] : A.sstride[3]));
}

__device__ __forceinline__ f32x4 wg_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// issue the loads of one iteration's operand rows (ET x 16 edges x [x0 | x1 | g | h]) into registers
template <int NP>
__device__ __forceinline__ void wg_load(const WgArgs& A, const int* __restrict__ U, int64_t e0, f32x4 (&st)[NP], int tid) {
    const int nsrc = U[0], PR = U[24], XP = U[25], GP = U[26], HP = U[27], ET = U[12];
    const int total = ET * 16 * PR;
    const float* __restrict__ b0 = wg_pick_src(A, U[1]) + U[3];
    const float* __restrict__ b1 = wg_pick_src(A, U[2]) + U[3];
    const int64_t st0 = wg_pick_stride(A, U[1]), st1 = wg_pick_stride(A, U[2]);
    const float* __restrict__ bg = A.g + U[7];
    const float* __restrict__ bh = U[9] ? A.h[1] : A.h[0];
    const int x1 = nsrc == 2 ? XP : 0;                         // pieces of source 1
    const int pg = XP + x1, ph = pg + GP, pe = ph + HP;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const int q = tid + n * WG_NT;
        if (n * WG_NT < total) {                               // uniform
            const int row = q / PR, pos = q - row * PR;
            int64_t e = e0 + row;
            e = e < A.rows ? e : A.rows - 1;                   // tail rows read a valid row; their products are zeroed (see wg_unit)
            const float* p;
            if (pos < XP) p = b0 + e * st0 + pos * 4;
            else if (pos < pg) p = b1 + e * st1 + (pos - XP) * 4;
            else if (pos < ph) p = bg + e * A.gstride + (pos - pg) * 4;
            else p = bh + e * A.hstride + (pos - ph) * 4;
            if (q < total && pos < pe) st[n] = *reinterpret_cast<const f32x4*>(p);
        }
    }
}

template <int NP>
__device__ __forceinline__ void wg_store(const int* __restrict__ U, float* __restrict__ buf, const f32x4 (&st)[NP], int tid) {
    const int nsrc = U[0], PR = U[24], XP = U[25], GP = U[26], HP = U[27], ET = U[12];
    const int total = ET * 16 * PR;
    const int pe = XP + (nsrc == 2 ? XP : 0) + GP + HP;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const int q = tid + n * WG_NT;
        if (n * WG_NT < total) {
            const int row = q / PR, pos = q - row * PR;
            if (q < total && pos < pe) *reinterpret_cast<f32x4*>(buf + q * 4) = st[n];     // row * RS + pos * 4 == q * 4 (RS = 4 PR)
        }
    }
}

template <int NC, int MAXT>
__device__ __forceinline__ void wg_unit(const WgArgs& A, const int* __restrict__ U, const float* __restrict__ Wg, const int* __restrict__ chtab,
                                        float* __restrict__ lds, int split, int nsplit) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nsrc = U[0], in_mulp = U[4], par = U[6], g_mulp = U[8], nrt = U[10], nrtp = U[11], ET = U[12], RS = U[13];
    const int xoff1 = U[14], goff = U[15], hoff = U[16], ntu = U[22], ntk = U[23];
    const int tau = wave & (nrtp - 1), et = wave / nrtp;
    const bool busy = tau < nrt && et < ET;
    const int64_t T = (A.rows + 15) >> 4;
    const int64_t NI = (T + ET - 1) / ET;
    const int64_t per = (NI + nsplit - 1) / nsplit;
    const int64_t it0 = (int64_t)split * per, it1 = (it0 + per) < NI ? (it0 + per) : NI;
    const int buf_floats = ET * 16 * RS;
    const int el = lane & 15, g = lane >> 4;

    const float* __restrict__ wt = Wg + U[17] + (busy ? tau : 0) * U[18];
    const int ksx = in_mulp >> 2, ksg = g_mulp >> 2, ksh = A.hidden >> 2;
    const int G1 = (ksx + 3) >> 2, G2 = (ksg + 3) >> 2, G3 = (ksh + 3) >> 2;
    const float* __restrict__ wW = wt + lane * 4;
    const float* __restrict__ wL = wW + nsrc * G1 * 256;
    const float* __restrict__ w3 = wL + G2 * 256;
    const float* __restrict__ wcf = wt + (nsrc * G1 + G2 + G3) * 256;
    float cfv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cfv[c] = wcf[c * 16 + el];
    const int ch = chtab[U[21] + (busy ? tau : 0) * 16 + el];
    float* __restrict__ gsp = U[9] ? A.gs[1] : A.gs[0];
    const int64_t gss = U[9] ? A.gsstride[1] : A.gsstride[0];

    f32x4 accW[2][MAXT], accL[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        accW[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        accW[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        accL[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 st[WG_PIECES_OF(NC)];
    if (it0 < it1) {
        wg_load(A, U, it0 * ET * 16, st, tid);
        wg_store(U, lds, st, tid);
    }
    __syncthreads();
#pragma unroll 1
    for (int64_t it = it0; it < it1; ++it) {
#ifdef WG_ABL_NOLOAD
        const float* __restrict__ cur = lds;
        float* __restrict__ nxt = lds + buf_floats;
#else
        const float* __restrict__ cur = lds + ((it - it0) & 1) * buf_floats;
        float* __restrict__ nxt = lds + (((it - it0) & 1) ^ 1) * buf_floats;
#endif
        const float* __restrict__ rowbase = cur + et * 16 * RS;
        const int64_t e_tile = (it * ET + et) * 16;
        // ---- the next iteration's rows start travelling (WG_EARLY_LOAD, measured no faster: vmcnt is in-order, so the first weight-fragment wait of phase 1 drains these loads too)
        const bool more = it + 1 < it1;
#if !defined(WG_ABL_NOLOAD) && defined(WG_EARLY_LOAD)
        if (more) wg_load(A, U, (it + 1) * ET * 16, st, tid);
#endif
        f32x4 mid[NC], bm[NC];
        if (busy) {
            // ---- phase 1: mid^T, B^T, s^T (A operand: lane (edge el, K-slot g) reads one dword of its edge's row)
            const float* __restrict__ xa = rowbase + el * RS + g;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                mid[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                bm[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef WG_ABL_NOP1
#pragma unroll 1
            for (int s = 0; s < nsrc; ++s) {
                const float* __restrict__ xs = xa + (s ? xoff1 : 0);
                const float* __restrict__ wf = wW + s * G1 * 256;
#pragma unroll 1
                for (int G = 0; G < G1; ++G) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wf + G * 256);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int step = 4 * G + q;
                        if (step < ksx) {
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                const int cl = par ? (NC - 1 - c) : c;
                                mid[c] = wg_mfma(xs[cl * in_mulp + 4 * step], w[q], mid[c]);
                            }
                        }
                    }
                }
            }
            {
                const float* __restrict__ ga = xa + goff;
#pragma unroll 1
                for (int G = 0; G < G2; ++G) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wL + G * 256);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int step = 4 * G + q;
                        if (step < ksg) {
#pragma unroll
                            for (int c = 0; c < NC; ++c) bm[c] = wg_mfma(ga[c * g_mulp + 4 * step], w[q], bm[c]);
                        }
                    }
                }
            }
            {
                const float* __restrict__ ha = xa + hoff;
#pragma unroll 1
                for (int G = 0; G < G3; ++G) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(w3 + G * 256);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * G + q < ksh) sv = wg_mfma(ha[4 * (4 * G + q)], w[q], sv);
                }
            }
#endif
            // ---- element-wise (lane (row el, g): edges 4 g + r): gs, T1 = s cf B (-> bm), T2 = s cf mid (-> mid)
            f32x4 gsr = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 a = mid[c] * cfv[c];
                gsr += a * bm[c];
                bm[c] = sv * cfv[c] * bm[c];
                mid[c] = sv * a;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t e = e_tile + 4 * g + r;
                const bool valid = e < A.rows;
#ifndef WG_ABL_NOGS
                if (valid && ch >= 0) gsp[e * gss + ch] = gsr[r];
#endif
                if (!valid) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        bm[c][r] = 0.f;
                        mid[c][r] = 0.f;
                    }
                }
            }
        }
#if !defined(WG_ABL_NOLOAD) && !defined(WG_EARLY_LOAD)
        if (more) wg_load(A, U, (it + 1) * ET * 16, st, tid);
#endif
#ifndef WG_ABL_NOP2
        if (busy) {
            // ---- phase 2: K = the 16 edges (K-step r: slot g <-> edge 4 g + r); A operand: lane (channel el, g) reads x[edge 4 g + r][channel]
            const float* __restrict__ xb = rowbase + (4 * g) * RS + el;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int cl = par ? (NC - 1 - c) : c;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s < nsrc) {
                        const float* __restrict__ xs = xb + (s ? xoff1 : 0) + cl * in_mulp;
#pragma unroll
                        for (int t = 0; t < MAXT; ++t) {
                            if (t < ntu) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) accW[s][t] = wg_mfma(xs[r * RS + 16 * t], bm[c][r], accW[s][t]);
                            }
                        }
                    }
                }
                const float* __restrict__ gb = xb + goff + c * g_mulp;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (t < ntk) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) accL[t] = wg_mfma(gb[r * RS + 16 * t], mid[c][r], accL[t]);
                    }
                }
            }
        }
#endif
#ifndef WG_ABL_NOLOAD
        if (more) wg_store(U, nxt, st, tid);
#endif
        __syncthreads();
    }
    // ---- accumulators: block (split, unit, edge-tile copy, row tile), fragment f, [row][channel]: lane (row el, g) holds channels 4 g .. 4 g + 3
    if (busy) {
        float* __restrict__ ab = A.acc + (int64_t)split * A.acc_split + U[19] + (et * nrt + tau) * U[20] + el * 16 + 4 * g;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
                if (s < nsrc && t < ntu) *reinterpret_cast<f32x4*>(ab + (s * ntu + t) * 256) = accW[s][t];
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < ntk) *reinterpret_cast<f32x4*>(ab + (nsrc * ntu + t) * 256) = accL[t];
    }
}

extern "C" __global__ void __launch_bounds__(WG_NT, 2)
tp_wgrad_kernel(const WgArgs A, const int* __restrict__ units, const float* __restrict__ weights, const int* __restrict__ chtab) {
    extern __shared__ float lds[];
    const int* __restrict__ U = units + (size_t)blockIdx.x * WG_UNIT_I32;
    const int split = blockIdx.y, nsplit = gridDim.y;
    switch (U[5]) {
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 1
        case 1: wg_unit<1, WG_MAXT_OF(1)>(A, U, weights, chtab, lds, split, nsplit); break;
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 3
        case 3: wg_unit<3, WG_MAXT_OF(3)>(A, U, weights, chtab, lds, split, nsplit); break;
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 5
        case 5: wg_unit<5, WG_MAXT_OF(5)>(A, U, weights, chtab, lds, split, nsplit); break;
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 7
        case 7: wg_unit<7, WG_MAXT_OF(7)>(A, U, weights, chtab, lds, split, nsplit); break;
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 9
        case 9: wg_unit<9, WG_MAXT_OF(9)>(A, U, weights, chtab, lds, split, nsplit); break;
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 11
        case 11: wg_unit<11, WG_MAXT_OF(11)>(A, U, weights, chtab, lds, split, nsplit); break;
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 13
        case 13: wg_unit<13, WG_MAXT_OF(13)>(A, U, weights, chtab, lds, split, nsplit); break;
#endif
        default: break;
    }
}

// C ABI (include/hamgnn_hip.h): see there for the argument meaning
extern "C" int hg_tp_wgrad(const float* const* src, const int64_t* src_stride, int nsrc_slots, const float* g, int64_t g_stride,
                           const float* h_node, const float* h_edge, int64_t h_stride, int hidden,
                           float* gs_node, int64_t gs_node_stride, float* gs_edge, int64_t gs_edge_stride,
                           float* acc, int64_t acc_floats, int nsplit, const int32_t* units, int nunits, const float* weights, const int32_t* chtab,
                           int lds_bytes, int64_t rows, void* stream) {
    if (rows <= 0 || nunits <= 0) return 0;
    if (nsplit < 1 || nsrc_slots > 4 || hidden % 16 || lds_bytes > 160 * 1024) return -1;
    WgArgs A;
    for (int i = 0; i < 4; ++i) {
        A.src[i] = i < nsrc_slots ? src[i] : nullptr;
        A.sstride[i] = i < nsrc_slots ? src_stride[i] : 0;
    }
    A.g = g;
    A.gstride = g_stride;
    A.h[0] = h_node;
    A.h[1] = h_edge ? h_edge : h_node;
    A.hstride = h_stride;
    A.gs[0] = gs_node;
    A.gs[1] = gs_edge ? gs_edge : gs_node;
    A.gsstride[0] = gs_node_stride;
    A.gsstride[1] = gs_edge ? gs_edge_stride : gs_node_stride;
    A.acc = acc;
    A.acc_split = acc_floats;
    A.rows = rows;
    A.hidden = hidden;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(tp_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
        attr_set = true;
    }
    hipLaunchKernelGGL(tp_wgrad_kernel, dim3(nunits, nsplit), dim3(WG_NT), lds_bytes, static_cast<hipStream_t>(stream), A, units, weights, chtab);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

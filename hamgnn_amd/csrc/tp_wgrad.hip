// tp_wgrad.hip -- fused WEIGHT gradients of the weighted tensor-product branches of a MessagePackBlock (the backward of
// /root/reference/hamgnn/nn/message_passing.py:191-231 with respect to tensor_product.weight, linear_scaler.linear_out.weight,
// linear_out.weight and -- through gs -- the radial weight generators).  Hand-written HIP for gfx950 (CDNA4), fp32 MFMA.
//
// Weight-stationary where the forward (csrc/tp_is.hip) is input-stationary: a workgroup owns one UNIT of plan.WgFused = up to four
// 16-row tiles of super-paths that read the SAME input irrep i of one branch (rows = (e3nn path, mid channel); the tiles may belong to
// different output irreps k) and a range of 16-edge tiles; one wave per row tile, its weights AND accumulators never leave the registers.
// Per edge tile, with EDGES as the MFMA M index (A operand = staged rows from LDS, B operand = weight fragments in registers):
//     mid^T[e, row] (per column c) = x[e, :, comp(c)] . W[row, :]        B^T[e, row] = g[e, :, col(c)] . L[row, :]        s^T[e, row] = h[e, :] . W3[:, row]
// The C fragments hold, for lane (row, g), the edges 4 g + r: exactly the B-operand layout of a K = 16-edges MFMA, so
//     g_W[u, row] += sum_e x[e, u, comp(c)] * (s cf B)[e, row, c]          g_L[w, row] += sum_e g[e, w, col(c)] * (s cf mid)[e, row, c]
// are issued straight from those registers (A operand = the same staged rows read along the edge axis; no transposition, nothing
// materialised).  gs[e, ch(row)] = sum_c cf mid B is written per edge (last radial layer / hidden-layer gradients: two library GEMMs).
// Every loop bound of the two MFMA phases is a template parameter <NC columns, NSRC sources, G1 / G2 = tiles of 16 input / output
// channels>: no branches between the LDS reads and the MFMAs (the first version guarded every tile at run time and ran at 17 % of the pipe).
// Staging: the next tile's rows travel through registers (float4 loads issued at the top of the iteration -- the MFMA phases issue no other
// load: vmcnt is in-order -- and written to the other LDS buffer at its end): one barrier per iteration.  LDS rows are
// [x source 0 | x source 1 | g spans | h] with stride == 4 (mod 64) floats: the dword reads of (16 edges x 4 K-slots) and of (4 edges x
// 16 channels) are both conflict-free.
// Partial sums of the splits (blockIdx.y) and of the edge-tile copies inside a workgroup go to separate accumulator blocks; the host adds
// them in a fixed order (bit-reproducible, no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WG_UNIT_I32 64
#define WG_NT 256
#define WG_NP 10                 // float4 pieces a thread holds in flight while the next tile is staged ...
#define WG_NP_OF(NC) ((NC) <= 9 ? WG_NP : 5)    // ... by column count of the wave (plan.wg_pieces_of_nc): 11 / 13 columns hold 88 / 104 product registers
#define WG_WREC 24               // first per-wave record
#define WG_WREC_I32 10

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

// issue the loads of one iteration's operand rows (ET x 16 edges x [x0 | x1 | g spans | h]) into registers.  16 / ET threads share a row
// (piece pos = j + (16 / ET) n of row tid / (16 / ET)): no division, one set of row pointers per thread and iteration
template <int NP>
__device__ __forceinline__ void wg_load(const WgArgs& A, const int* __restrict__ U, int64_t e0, f32x4 (&st)[NP], int tid) {
    const int nsrc = U[0], XP = U[5], HP = U[11], ET = U[8];
    const int sh = ET == 1 ? 4 : (ET == 2 ? 3 : 2), tpr = 1 << sh;
    const int row = tid >> sh, j = tid & (tpr - 1);
    int64_t e = e0 + row;
    e = e < A.rows ? e : A.rows - 1;                           // tail rows read a valid row; their products are zeroed (see wg_wave)
    const float* __restrict__ r0 = wg_pick_src(A, U[1]) + U[3] + e * wg_pick_stride(A, U[1]);
    const float* __restrict__ r1 = wg_pick_src(A, U[2]) + U[3] + e * wg_pick_stride(A, U[2]);
    const float* __restrict__ rg = A.g + e * A.gstride;
    const float* __restrict__ rh = (U[7] ? A.h[1] : A.h[0]) + e * A.hstride;
    const int p1 = XP, p2 = nsrc == 2 ? 2 * XP : XP;           // [0, p1) source 0, [p1, p2) source 1
    const int q0 = p2 + U[17], q1 = q0 + U[19], q2 = q1 + U[21], q3 = q2 + U[23];     // ends of the (<= 4) gradient spans
    const int pe = q3 + HP;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const int pos = j + n * tpr;
        if (n * tpr < pe) {                                    // uniform
            const float* p;
            if (pos < p1) p = r0 + pos * 4;
            else if (pos < p2) p = r1 + (pos - p1) * 4;
            else if (pos < q0) p = rg + U[16] + (pos - p2) * 4;
            else if (pos < q1) p = rg + U[18] + (pos - q0) * 4;
            else if (pos < q2) p = rg + U[20] + (pos - q1) * 4;
            else if (pos < q3) p = rg + U[22] + (pos - q2) * 4;
            else p = rh + (pos - q3) * 4;
            if (pos < pe) st[n] = *reinterpret_cast<const f32x4*>(p);
        }
    }
}

template <int NP>
__device__ __forceinline__ void wg_store(const int* __restrict__ U, float* __restrict__ buf, const f32x4 (&st)[NP], int tid) {
    const int ET = U[8], RS = U[9];
    const int sh = ET == 1 ? 4 : (ET == 2 ? 3 : 2), tpr = 1 << sh;
    const int row = tid >> sh, j = tid & (tpr - 1);
    const int pe = (U[0] == 2 ? 2 : 1) * U[5] + U[17] + U[19] + U[21] + U[23] + U[11];
    float* __restrict__ d = buf + row * RS + j * 4;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const int pos = j + n * tpr;
        if (n * tpr < pe) {
            if (pos < pe) *reinterpret_cast<f32x4*>(d + n * tpr * 4) = st[n];
        }
    }
}

struct WgRange {
    int64_t it0, it1;
};
__device__ __forceinline__ WgRange wg_range(const WgArgs& A, int ET, int split, int nsplit) {
    const int64_t T = (A.rows + 15) >> 4;
    const int64_t NI = (T + ET - 1) / ET;
    const int64_t per = (NI + nsplit - 1) / nsplit;
    WgRange r;
    r.it0 = (int64_t)split * per;
    r.it1 = (r.it0 + per) < NI ? (r.it0 + per) : NI;
    return r;
}

// a wave without a row tile: staging and barriers only
__device__ __forceinline__ void wg_idle(const WgArgs& A, const int* __restrict__ U, float* __restrict__ lds, int split, int nsplit) {
    const int tid = threadIdx.x, ET = U[8];
    const WgRange R = wg_range(A, ET, split, nsplit);
    const int buf_floats = ET * 16 * U[9];
    f32x4 st[WG_NP];
    if (R.it0 < R.it1) {
        wg_load(A, U, R.it0 * ET * 16, st, tid);
        wg_store(U, lds, st, tid);
    }
    __syncthreads();
#pragma unroll 1
    for (int64_t it = R.it0; it < R.it1; ++it) {
        float* __restrict__ nxt = lds + (((it - R.it0) & 1) ^ 1) * buf_floats;
        const bool more = it + 1 < R.it1;
        if (more) {
            wg_load(A, U, (it + 1) * ET * 16, st, tid);
            wg_store(U, nxt, st, tid);
        }
        __syncthreads();
    }
}

// Every LDS operand of the MFMA phases has ONE consumer, and the scheduler emits read -> s_waitcnt lgkmcnt(0) -> v_mfma per operand: an LDS
// round trip per MFMA, in all four waves of a workgroup at once (they run the phases in lock step).  WG_SGB(cond, n): "n LDS reads, then
// n MFMAs" for the instructions just written (ISA audit, profiles/r03_wgrad.md); -DWG_NO_SGB restores the compiler's order.
#ifndef WG_NO_SGB
#define WG_SGB(cond, n) if (cond) { __builtin_amdgcn_sched_group_barrier(0x100, (n), 0); __builtin_amdgcn_sched_group_barrier(0x008, (n), 0); }
#else
#define WG_SGB(cond, n)
#endif
template <int NC, int NSRC, int G1, int G2>
__device__ __forceinline__ void wg_wave(const WgArgs& A, const int* __restrict__ U, const int* __restrict__ Wr, const float* __restrict__ Wg,
                                        const int* __restrict__ chtab, float* __restrict__ lds, int split, int nsplit) {
    constexpr int G3 = 4;                                      // hidden = 64 (checked on the host)
    const int tid = threadIdx.x, lane = tid & 63;
    const int in_mulp = U[4], ET = U[8], RS = U[9];
    const int et = Wr[1], par = Wr[3], xc0 = Wr[4], goff = Wr[5], g_mulp = Wr[6];
    const int xoff1 = 4 * U[5];
    const int hoff = 4 * ((NSRC == 2 ? 2 : 1) * U[5] + U[17] + U[19] + U[21] + U[23]);
    const WgRange R = wg_range(A, ET, split, nsplit);
    const int buf_floats = ET * 16 * RS;
    const int el = lane & 15, g = lane >> 4;
    const int ksx = in_mulp >> 2, ksg = g_mulp >> 2;

    // ---- the row tile's weights: B-operand fragments, resident for the whole edge range
    const float* __restrict__ wt = Wg + Wr[7] + lane * 4;
    f32x4 fW[NSRC][G1], fL[G2], f3[G3];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int G = 0; G < G1; ++G) fW[s][G] = *reinterpret_cast<const f32x4*>(wt + (s * G1 + G) * 256);
#pragma unroll
    for (int G = 0; G < G2; ++G) fL[G] = *reinterpret_cast<const f32x4*>(wt + (NSRC * G1 + G) * 256);
#pragma unroll
    for (int G = 0; G < G3; ++G) f3[G] = *reinterpret_cast<const f32x4*>(wt + (NSRC * G1 + G2 + G) * 256);
    const float* __restrict__ wcf = Wg + Wr[7] + (NSRC * G1 + G2 + G3) * 256;
    float cfv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cfv[c] = wcf[c * 16 + el];
    const int ch = chtab[Wr[9] + el];
    float* __restrict__ gsp = U[7] ? A.gs[1] : A.gs[0];
    const int64_t gss = U[7] ? A.gsstride[1] : A.gsstride[0];

    f32x4 accW[NSRC][G1], accL[G2];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int t = 0; t < G1; ++t) accW[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < G2; ++t) accL[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 st[WG_NP_OF(NC)];
    if (R.it0 < R.it1) {
        wg_load(A, U, R.it0 * ET * 16, st, tid);
        wg_store(U, lds, st, tid);
    }
    __syncthreads();
#pragma unroll 1
    for (int64_t it = R.it0; it < R.it1; ++it) {
#ifdef WG_ABL_NOLOAD
        const float* __restrict__ cur = lds;
        float* __restrict__ nxt = lds + buf_floats;
#else
        const float* __restrict__ cur = lds + ((it - R.it0) & 1) * buf_floats;
        float* __restrict__ nxt = lds + (((it - R.it0) & 1) ^ 1) * buf_floats;
#endif
        const float* __restrict__ rowbase = cur + et * 16 * RS;
        const int64_t e_tile = (it * ET + et) * 16;
        const bool more = it + 1 < R.it1;
        // ---- the next iteration's rows start travelling: neither MFMA phase issues a load (the weights are resident), so they have the whole
        // iteration to land (vmcnt is in-order: any later load would have to wait for these)
#if !defined(WG_ABL_NOLOAD) && !defined(WG_LATE_LOAD)
        if (more) wg_load(A, U, (it + 1) * ET * 16, st, tid);
#endif
        f32x4 mid[NC], bm[NC];
        f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            // ---- phase 1: mid^T, B^T, s^T (A operand: lane (edge el, K-slot g) reads one dword of its edge's row).  K-steps of the last
            // channel tile beyond the block's channels are skipped (uniform branch around whole steps)
            const float* __restrict__ xa = rowbase + el * RS + g;
            f32x4 mid2 = f32x4{0.f, 0.f, 0.f, 0.f}, bm2 = f32x4{0.f, 0.f, 0.f, 0.f};    // NC == 1: two accumulators per product (no back-to-back dependent MFMAs)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                mid[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                bm[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#ifndef WG_ABL_NOP1
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const float* __restrict__ xs = xa + (s ? xoff1 : 0) + xc0;
#pragma unroll
                for (int G = 0; G < G1; ++G) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int step = 4 * G + q;
                        if (G + 1 < G1 || q == 0 || step < ksx) {        // only the last tile's trailing steps are conditional
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                const int cl = par ? (NC - 1 - c) : c;
                                const float a = xs[cl * in_mulp + 4 * step];
                                if (NC == 1 && (q & 1)) mid2 = wg_mfma(a, fW[s][G][q], mid2);
                                else mid[c] = wg_mfma(a, fW[s][G][q], mid[c]);
                            }
                            WG_SGB(NC > 1, NC)
                        }
                    }
                    WG_SGB(NC == 1 && G + 1 < G1, 4)
                }
            }
            {
                const float* __restrict__ ga = xa + goff;
#pragma unroll
                for (int G = 0; G < G2; ++G) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int step = 4 * G + q;
                        if (G + 1 < G2 || q == 0 || step < ksg) {
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                const float a = ga[c * g_mulp + 4 * step];
                                if (NC == 1 && (q & 1)) bm2 = wg_mfma(a, fL[G][q], bm2);
                                else bm[c] = wg_mfma(a, fL[G][q], bm[c]);
                            }
                            WG_SGB(NC > 1, NC)
                        }
                    }
                    WG_SGB(NC == 1 && G + 1 < G2, 4)
                }
            }
            {
                const float* __restrict__ ha = xa + hoff;
                f32x4 sv2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int G = 0; G < G3; ++G) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q & 1) sv2 = wg_mfma(ha[4 * (4 * G + q)], f3[G][q], sv2);
                        else sv = wg_mfma(ha[4 * (4 * G + q)], f3[G][q], sv);
                    }
                    WG_SGB(true, 4)
                }
                sv += sv2;
            }
#endif
            if (NC == 1) {
                mid[0] += mid2;
                bm[0] += bm2;
            }
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
#if !defined(WG_ABL_NOLOAD) && defined(WG_LATE_LOAD)
        if (more) wg_load(A, U, (it + 1) * ET * 16, st, tid);
#endif
#ifndef WG_ABL_NOP2
        {
            // ---- phase 2: K = the 16 edges (K-step r: slot g <-> edge 4 g + r); A operand: lane (channel el, g) reads x[edge 4 g + r][channel]
            const float* __restrict__ xb = rowbase + (4 * g) * RS + el;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int cl = par ? (NC - 1 - c) : c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int s = 0; s < NSRC; ++s) {
                        const float* __restrict__ xs = xb + (s ? xoff1 : 0) + xc0 + cl * in_mulp + r * RS;
#pragma unroll
                        for (int t = 0; t < G1; ++t) accW[s][t] = wg_mfma(xs[16 * t], bm[c][r], accW[s][t]);
                    }
                    const float* __restrict__ gb = xb + goff + c * g_mulp + r * RS;
#pragma unroll
                    for (int t = 0; t < G2; ++t) accL[t] = wg_mfma(gb[16 * t], mid[c][r], accL[t]);
                    WG_SGB(true, NSRC * G1 + G2)
                }
            }
        }
#endif
#ifndef WG_ABL_NOLOAD
        if (more) wg_store(U, nxt, st, tid);
#endif
        __syncthreads();
    }
    // ---- accumulators: block (split, unit, wave), fragment f, [row][channel]: lane (row el, g) holds channels 4 g .. 4 g + 3
    float* __restrict__ ab = A.acc + (int64_t)split * A.acc_split + Wr[8] + el * 16 + 4 * g;
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int t = 0; t < G1; ++t) *reinterpret_cast<f32x4*>(ab + (s * G1 + t) * 256) = accW[s][t];
#pragma unroll
    for (int t = 0; t < G2; ++t) *reinterpret_cast<f32x4*>(ab + (NSRC * G1 + t) * 256) = accL[t];
}

// instantiated shapes (plan.wg_shape_ok): NC = 1: G1, G2 in 1..4; NC = 3, 5, 7: G1, G2 in 1..2; NC = 9, 11, 13: G1 = G2 = 1; NSRC in 1..2.
// code = ((NC >> 1) * 2 + (NSRC - 1)) * 16 + (G1 - 1) * 4 + (G2 - 1)
#define WG_CODE(NC, NSRC, G1, G2) ((((NC) >> 1) * 2 + ((NSRC) - 1)) * 16 + ((G1) - 1) * 4 + ((G2) - 1))
#define WG_CASE(NC, NSRC, G1, G2) \
    case WG_CODE(NC, NSRC, G1, G2): wg_wave<NC, NSRC, G1, G2>(A, U, Wr, weights, chtab, lds, split, nsplit); break;
#define WG_CASES_G(NC, NSRC, G1) WG_CASE(NC, NSRC, G1, 1) WG_CASE(NC, NSRC, G1, 2)
#define WG_CASES_G4(NC, NSRC, G1) WG_CASE(NC, NSRC, G1, 1) WG_CASE(NC, NSRC, G1, 2) WG_CASE(NC, NSRC, G1, 3) WG_CASE(NC, NSRC, G1, 4)
#define WG_CASES_MID(NC) WG_CASES_G(NC, 1, 1) WG_CASES_G(NC, 1, 2) WG_CASES_G(NC, 2, 1) WG_CASES_G(NC, 2, 2)
#define WG_CASES_HI(NC) WG_CASE(NC, 1, 1, 1) WG_CASE(NC, 2, 1, 1)

extern "C" __global__ void __launch_bounds__(WG_NT, 2)
tp_wgrad_kernel(const WgArgs A, const int* __restrict__ units, const float* __restrict__ weights, const int* __restrict__ chtab) {
    extern __shared__ float lds[];
    const int* __restrict__ U = units + (size_t)blockIdx.x * WG_UNIT_I32;
    const int split = blockIdx.y, nsplit = gridDim.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int* __restrict__ Wr = U + WG_WREC + wave * WG_WREC_I32;
    if (!Wr[0]) {
        wg_idle(A, U, lds, split, nsplit);
        return;
    }
    const int nc = Wr[2], nsrc = U[0], g1 = U[12], g2 = (Wr[6] + 15) >> 4;
    switch (WG_CODE(nc, nsrc, g1, g2)) {
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 1
        WG_CASES_G4(1, 1, 1) WG_CASES_G4(1, 1, 2) WG_CASES_G4(1, 1, 3) WG_CASES_G4(1, 1, 4)
        WG_CASES_G4(1, 2, 1) WG_CASES_G4(1, 2, 2) WG_CASES_G4(1, 2, 3) WG_CASES_G4(1, 2, 4)
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 3
        WG_CASES_MID(3)
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 5
        WG_CASES_MID(5)
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 7
        WG_CASES_MID(7)
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 9
        WG_CASES_HI(9)
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 11
        WG_CASES_HI(11)
#endif
#if !defined(WG_ONLY_NC) || WG_ONLY_NC == 13
        WG_CASES_HI(13)
#endif
        default: wg_idle(A, U, lds, split, nsplit); break;     // (the planner never emits another shape)
    }
}

// C ABI (include/hamgnn_hip.h): see there for the argument meaning
extern "C" int hg_tp_wgrad(const float* const* src, const int64_t* src_stride, int nsrc_slots, const float* g, int64_t g_stride,
                           const float* h_node, const float* h_edge, int64_t h_stride, int hidden,
                           float* gs_node, int64_t gs_node_stride, float* gs_edge, int64_t gs_edge_stride,
                           float* acc, int64_t acc_floats, int nsplit, const int32_t* units, int nunits, const float* weights, const int32_t* chtab,
                           int lds_bytes, int64_t rows, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0 || nunits <= 0) return 0;
    if (nsplit < 1 || nsrc_slots > 4 || lds_bytes > 160 * 1024) return hg_fail(-2, "hg_tp_wgrad: bad split / source count / LDS size");
    if (hidden != 64) return hg_fail(-2, "hg_tp_wgrad: the kernel is built for a 64-wide last hidden layer of the radial MLP");
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
    static unsigned char lds_attr_done[HG_MAX_DEVICES];       // once per device (not a stream operation: illegal during graph capture)
    if (int rc = hg_lds_attr_once(lds_attr_done, dev_guard.dev, (const void*)tp_wgrad_kernel, 160 * 1024)) return rc;
    hipLaunchKernelGGL(tp_wgrad_kernel, dim3(nunits, nsplit), dim3(WG_NT), lds_bytes, static_cast<hipStream_t>(stream), A, units, weights, chtab);
    return hg_check_launch("hg_tp_wgrad");
}

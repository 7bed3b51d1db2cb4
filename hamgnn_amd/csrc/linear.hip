// linear.hip -- o3.Linear on planar rows as ONE streaming pass (gfx950).  Hand-written HIP.
//
// Reference: every e3nn o3.Linear of the hot path (hamgnn/nn/interaction_blocks.py:332-358 ResidualBlock linear1 / linear2,
// nn/convolution.py:127 skip, nn/interaction_blocks.py:141-152 linear_up_*, models/hamgnn_output.py:49-58 HamLayer.linear_transform).
// An o3.Linear is block diagonal over the irreps: for every output irrep k, Y[(row, a), :] = X_i[(row, a), :] W_ik (summed over the
// matching input irreps i), where a "pair-row" (row, a) -- one component of one irrep of one feature row -- is a run of contiguous
// floats in the planar layout.  Per edge row that is ~40 kFLOP on 7 KB of traffic: by the roofline the op is an HBM stream.
// No LDS, no barriers: a wave unit = (<= 64 output channels of one irrep block, one component a) on 2 x 16 rows; every input float is
// loaded once per unit (float4 per lane straight into the MFMA B operand, K permuted to match: plan._frag_A), the weight fragments
// (<= 16 KB per unit) come from L1 / L2, the C fragment of  W^T x X^T  is a float4 of four consecutive output channels of one
// pair-row: stored as such, with up to two residual rows added on the way (ResidualBlock: x + Lin2(...) [+ skip]).
// Measured (profiles/r02_linear.md, 822 350 rows): 877 -> 1012: 3.3 ms, 877 -> 877 + residual: 4.1 ms (the segment-stationary program
// kernel: 4.4 / 4.3 ms) = 1.9 - 2.1 TB/s.  HBM traffic is the algorithmic 6 - 9 GB (FETCH / WRITE_SIZE), but the kernel is NOT at the HBM
// roofline: with all loads served from cache and the stores dropped it still takes 2.1 ms -- 37 MFMAs per unit against ~470 other
// VALU instructions (64-bit addressing, predicates) and a three-deep scalar table chain per unit at two waves per SIMD.  The next
// step is rows staged whole through LDS by a workgroup (contiguous 3.5 KB reads, 32-bit LDS addressing); not built.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

typedef float ln_f4 __attribute__((ext_vector_type(4)));
#ifndef LN_TPW
#define LN_TPW 2                  // 16-row tiles a wave works on at once (measured: 1 -> 3.9 / 4.5 ms, 2 -> 3.3 / 4.1, 4 -> 3.4 / 4.9)
#endif

struct LinArgs {
    const float* x;
    int64_t xs;
    float* y;
    int64_t ys;
    const float* res[2];
    int64_t rs[2];
    int64_t rows;
};

__global__ __launch_bounds__(256) void linear_planar_kernel(const LinArgs A, const int* __restrict__ g_items, const int* __restrict__ g_units,
                                                            const int* __restrict__ g_paths, const float* __restrict__ g_W, int nitems) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, n = lane & 15;
    // ROW-BLOCK-major work order: a block of 16 LN_TPW rows is worked on by the `nitems` wave units (<= 64 output channels of one irrep
    // block, one component a) of ceil(nitems / 4) workgroups that all run on ONE XCD (the dispatcher deals consecutive workgroups
    // round-robin to the 8 XCDs: workgroup b -> XCD b % 8) at about the same time, so a feature row is pulled from HBM once into one L2.
    const int wgs_per_block = (nitems + 3) >> 2;
    const int64_t j = blockIdx.x >> 3;
    const int64_t rb = (j / wgs_per_block) * 8 + (blockIdx.x & 7);
    const int item = (int)(j % wgs_per_block) * 4 + wave;
    const int64_t r0 = rb * (16 * LN_TPW);
    if (item >= nitems || r0 >= A.rows) return;
    const int acomp = g_items[2 * item + 1];
    const int* __restrict__ U = g_units + g_items[2 * item] * 8;
    const int out_off = U[0], out_mulp = U[1], rtm = U[2], nstore = U[3], pb = U[4], pe = U[5];
    // LN_TPW tiles of 16 rows at once: their input float4s are requested together and every weight fragment is loaded once for all
    int64_t row[LN_TPW];
    bool valid[LN_TPW];
    ln_f4 acc[LN_TPW][4];
#pragma unroll
    for (int t = 0; t < LN_TPW; ++t) {
        const int64_t r = r0 + 16 * t + n;
        valid[t] = r < A.rows;
        row[t] = valid[t] ? r : A.rows - 1;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[t][rt] = (ln_f4){0.f, 0.f, 0.f, 0.f};
    }
    for (int p = pb; p < pe; ++p) {
        const int* __restrict__ Pp = g_paths + p * 4;
        const int in_off = Pp[0], in_mulp = Pp[1], ngrp = Pp[2];
        const ln_f4* __restrict__ wf = reinterpret_cast<const ln_f4*>(g_W + Pp[3]) + lane;
        const float* __restrict__ xin[LN_TPW];
#pragma unroll
        for (int t = 0; t < LN_TPW; ++t) xin[t] = A.x + row[t] * A.xs + in_off + acomp * in_mulp + 4 * g;
        ln_f4 b[LN_TPW], bn[LN_TPW];
        const bool k0 = 4 * g < in_mulp;
#pragma unroll
        for (int t = 0; t < LN_TPW; ++t) bn[t] = k0 ? *reinterpret_cast<const ln_f4*>(xin[t]) : (ln_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int G = 0; G < ngrp; ++G) {
#pragma unroll
            for (int t = 0; t < LN_TPW; ++t) b[t] = bn[t];
            if (G + 1 < ngrp) {                                // next K group in flight under this group's MFMAs
                const bool kn = (16 * (G + 1) + 4 * g) < in_mulp;
#pragma unroll
                for (int t = 0; t < LN_TPW; ++t) bn[t] = kn ? *reinterpret_cast<const ln_f4*>(xin[t] + 16 * (G + 1)) : (ln_f4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < rtm) {
                    const ln_f4 av = wf[(G * rtm + rt) * 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int t = 0; t < LN_TPW; ++t) acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], b[t][q], acc[t][rt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < LN_TPW; ++t) {
        if (valid[t]) {
            const int64_t col = (int64_t)out_off + (int64_t)acomp * out_mulp + 4 * g;
            float* __restrict__ yo = A.y + row[t] * A.ys + col;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (rt < rtm && 16 * rt + 4 * g < nstore) {
                    ln_f4 v = acc[t][rt];
                    if (A.res[0]) v += *reinterpret_cast<const ln_f4*>(A.res[0] + row[t] * A.rs[0] + col + 16 * rt);
                    if (A.res[1]) v += *reinterpret_cast<const ln_f4*>(A.res[1] + row[t] * A.rs[1] + col + 16 * rt);
                    *reinterpret_cast<ln_f4*>(yo + 16 * rt) = v;
                }
            }
        }
    }
}

extern "C" int hg_linear_planar(const float* x, int64_t x_stride, const int32_t* items, int nitems,
                                const int32_t* units, const int32_t* paths, const float* weights, const float* res0, int64_t res0_stride,
                                const float* res1, int64_t res1_stride, int64_t rows, float* y, int64_t y_stride, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nitems < 1) return hg_fail(-2, "hg_linear_planar: no work items");
    if ((x_stride & 3) || (y_stride & 3) || (res0 && (res0_stride & 3)) || (res1 && (res1_stride & 3)) ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res0) | reinterpret_cast<uintptr_t>(res1) |
          reinterpret_cast<uintptr_t>(weights)) & 15))
        return hg_fail(-2, "hg_linear_planar: rows must be multiples of 4 floats and 16-byte aligned (planar rows are)");
    LinArgs A;
    A.x = x, A.xs = x_stride, A.y = y, A.ys = y_stride;
    A.res[0] = res0, A.res[1] = res1, A.rs[0] = res0_stride, A.rs[1] = res1_stride;
    A.rows = rows;
    const int64_t row_blocks = (rows + 16 * LN_TPW - 1) / (16 * LN_TPW);
    const int64_t blocks = ((row_blocks + 7) / 8) * 8 * ((nitems + 3) / 4);      // 8 XCDs x row blocks per XCD x workgroups per row block
    if (blocks > 0x7fffffffLL) return hg_fail(-2, "hg_linear_planar: too many rows for one launch");
    linear_planar_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(A, items, units, paths, weights, nitems);
    return hg_check_launch("hg_linear_planar");
}

// block_gemm.hip -- many SMALL independent matrix products in one launch:  C_u = scale_u * op(A_u) @ op(B_u)  for the units of a table.
// What it replaces: the per-irrep products of a MessagePackBlock's two trailing Linears, L'_k = linear_scaler_k @ linear_out_k / sqrt(mul_k)
// (/root/reference/hamgnn/nn/message_passing.py:122-130, tensor_products.py:118-140: the reference applies the two Linears one after the other),
// evaluated once per optimiser step by the device-side repack (hamgnn_amd/repack.py:mp_sources) and, transposed, by the backward of those two
// Linears (hamgnn_amd/backward_mp.py:TPWeightGrad.finish) -- 13 output irreps x 2 branches x 7 blocks = 182 (repack) + 364 (backward) library GEMMs
// of [<= 832, <= 64] x [<= 64, <= 64] per training step, each a launch of a few microseconds.  fp32 operands, fp64 accumulation (the host
// packer rounds once, from float64), fp32 or fp64 result.  Plain FMA on an LDS-tiled 64 x 64 output tile per workgroup: these are kFLOP-sized.
// Hand-written HIP for gfx950 (CDNA4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

#define BG_UNIT_I32 12           // {a_off, a_ld, a_trans, b_off, b_ld, b_trans, c_off, c_ld, M, N, K, scale (float bits)}
#define BG_T 64                  // output tile
#define BG_K 16                  // K chunk

template <typename OUT>
__global__ __launch_bounds__(256) void block_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, OUT* __restrict__ C,
                                                        const int* __restrict__ units) {
    const int* __restrict__ U = units + (size_t)blockIdx.x * BG_UNIT_I32;
    const int a_off = U[0], a_ld = U[1], a_tr = U[2], b_off = U[3], b_ld = U[4], b_tr = U[5], c_off = U[6], c_ld = U[7], M = U[8], N = U[9], K = U[10];
    const double scale = (double)__int_as_float(U[11]);
    const int tn = (N + BG_T - 1) / BG_T, tm = (M + BG_T - 1) / BG_T;
    if ((int)blockIdx.y >= tm * tn) return;
    const int m0 = ((int)blockIdx.y / tn) * BG_T, n0 = ((int)blockIdx.y % tn) * BG_T;
    __shared__ float sa[BG_K][BG_T + 1], sb[BG_K][BG_T + 1];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = 0; k0 < K; k0 += BG_K) {
        for (int i = tid; i < BG_K * BG_T; i += 256) {
            const int kk = i / BG_T, mm = i - kk * BG_T;       // element (row / column mm of the tile, K index kk)
            const int k = k0 + kk;
            const int m = m0 + mm, n = n0 + mm;
            sa[kk][mm] = (k < K && m < M) ? A[a_off + (a_tr ? (int64_t)k * a_ld + m : (int64_t)m * a_ld + k)] : 0.f;
            sb[kk][mm] = (k < K && n < N) ? B[b_off + (b_tr ? (int64_t)n * b_ld + k : (int64_t)k * b_ld + n)] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BG_K; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = (double)sa[kk][4 * ty + i];
                b[i] = (double)sb[kk][4 * tx + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * ty + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 4 * tx + j;
            if (n < N) C[c_off + (int64_t)m * c_ld + n] = (OUT)(scale * acc[i][j]);
        }
    }
}

// C ABI (include/hamgnn_hip.h)
extern "C" int hg_block_gemm(const float* a, const float* b, void* c, int c_is_double, const int32_t* units, int nunits, int max_tiles, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (nunits <= 0) return 0;
    if (max_tiles <= 0 || max_tiles > 65535) return hg_fail(-2, "hg_block_gemm: bad tile count");
    const dim3 grid((unsigned)nunits, (unsigned)max_tiles);
    if (c_is_double) hipLaunchKernelGGL(block_gemm_kernel<double>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, b, static_cast<double*>(c), units);
    else hipLaunchKernelGGL(block_gemm_kernel<float>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, b, static_cast<float*>(c), units);
    return hg_check_launch("hg_block_gemm");
}

// linear_wgrad.hip -- weight gradient of an o3.Linear on planar rows: ALL its paths in one launch (gfx950).  Hand-written HIP.
//
// Reference: what torch.autograd computes for the weight of every e3nn o3.Linear of the hot path when the reference trains
// (hamgnn/models/Model.py:150-196; the Linears of nn/interaction_blocks.py:332-358, nn/convolution.py:127, models/hamgnn_output.py:49-58).
// For a path (input irrep i -> output irrep k of the same (l, p)) the gradient is gW[u, v] = sum over (row, component a) of
// x[row, i, a, u] * g[row, k, a, v]: a GEMM with a tiny [mul_i x mul_k] result and a reduction length of rows x (2 l + 1).  r3 issued one library
// GEMM per path on strided copies of the planar blocks: ~290 GEMMs + ~570 copy launches per training step, 37 of its 154 ms -- library GEMMs
// with a 64 x 64 result have nothing to parallelise over but K, which the library does not split.
// Here: grid = (tile unit, row chunk).  A unit = 16 input channels x up to 64 output channels of one path; a workgroup's four waves take the
// chunk's rows four at a time (K = 4 rows per MFMA, one component at a time), the A operand (16 channels of x, 64 contiguous bytes per row)
// is loaded once per K-step and feeds up to four MFMAs.  Every (unit, chunk) writes its partial [16 x 64] block to scratch; the host adds the
// chunks in a fixed order (torch.sum over the chunk axis): deterministic, no float atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

typedef float lw_f4 __attribute__((ext_vector_type(4)));
#define LW_ROWS 1024            // rows of a chunk (hamgnn_amd/ops.py:LW_ROWS)

// unit record, int32[8]: {x_off, x_mulp, g_off, g_mulp, n = 2 l + 1, u0 (first input channel), v0 (first output channel), nv (output channels, <= 64)}
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ x, int64_t xs, const float* __restrict__ g, int64_t gs, int64_t rows,
                                                           const int* __restrict__ units, float* __restrict__ part, int nunits) {
    __shared__ lw_f4 red[3][4][64];                            // waves 1..3 hand their accumulators to wave 0
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;
    const int* __restrict__ U = units + blockIdx.x * 8;
    const int x_off = U[0], x_mulp = U[1], g_off = U[2], g_mulp = U[3], n = U[4], u0 = U[5], v0 = U[6], nv = U[7];
    const int64_t r0 = (int64_t)blockIdx.y * LW_ROWS;
    const int64_t r1 = r0 + LW_ROWS < rows ? r0 + LW_ROWS : rows;
    const int nt = (nv + 15) >> 4;                             // 16-channel tiles of the output block (1..4)
    const bool ua = u0 + i < x_mulp;                           // (channels beyond the padded block: another irrep's floats -> masked; padding slots hold 0)
    bool va[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) va[t] = t < nt && v0 + 16 * t + i < g_mulp;
    lw_f4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (lw_f4){0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ xb = x + x_off + u0 + i;
    const float* __restrict__ gb = g + g_off + v0 + i;
    // K-steps: (4 consecutive rows) x component; the four waves interleave the row groups
#pragma unroll 1
    for (int64_t r = r0 + 4 * wave; r < r1; r += 16) {
        const int64_t row = r + kk;
        const bool rv = row < r1;
        const float* __restrict__ xr = xb + (rv ? row : r1 - 1) * xs;
        const float* __restrict__ gr = gb + (rv ? row : r1 - 1) * gs;
#pragma unroll 1
        for (int a = 0; a < n; ++a) {
            const float av = (rv && ua) ? xr[a * x_mulp] : 0.f;
            float bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = (rv && va[t]) ? gr[a * g_mulp + 16 * t] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
        }
    }
    if (wave) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += red[w][t][lane];          // fixed order
        // C fragment: column j = lane & 15 (output channel), rows 4 (lane >> 4) + r (input channel): block [16][64] per (chunk, unit)
        float* __restrict__ o = part + ((int64_t)blockIdx.y * nunits + blockIdx.x) * 1024;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(4 * kk + r) * 64 + 16 * t + i] = acc[t][r];
    }
}

// C ABI (include/hamgnn_hip.h)
extern "C" int hg_linear_wgrad(const float* x, int64_t x_stride, const float* g, int64_t g_stride, int64_t rows, const int32_t* units, int nunits,
                               float* partial, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0 || nunits <= 0) return 0;
    const int64_t nchunk = (rows + LW_ROWS - 1) / LW_ROWS;
    if (nchunk > 65535) return hg_fail(-2, "hg_linear_wgrad: more than 65535 row chunks (split the rows on the host)");
    linear_wgrad_kernel<<<dim3((unsigned)nunits, (unsigned)nchunk), 256, 0, (hipStream_t)stream>>>(x, x_stride, g, g_stride, rows, units, partial, nunits);
    return hg_check_launch("hg_linear_wgrad");
}

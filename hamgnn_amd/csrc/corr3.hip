// corr3.hip -- the nu = 3 term of the MACE symmetric contraction of a CorrProductBlock built with `correlation: 3`
// (/root/reference/hamgnn/nn/interaction_blocks.py:234-260 -> toolbox/mace/modules/symmetric_contraction.py:148-230: the main einsum
// "[w] x v i k, ekc, bci, be -> bc [w] x v" whose two open component indices the following nu = 2 / nu = 1 steps contract with x again;
// U_matrix_real of toolbox/mace/tools/cg.py:16-131 for three factors).  ADDED onto the rows that hg_sym_contraction (csrc/head.hip) wrote for
// nu <= 2:
//     out[n, o, c] += sum_{(x, i, j, kap, v) in ent3[o]} v W3[z_n, kap, c] h[n, x, c] h[n, i, c] h[n, j, c]
// with U_3 given sparsely by plan.py:sym_contraction_tables (entries in the reference's path order, CSR rows per output element o).
// Node-level and optional (correlation 3 is not the reference's default): one workgroup per node, the node's hidden components in LDS, a lane
// owns one (output element, channel) and walks its entry list in order -- a fixed summation order, no atomics.  HBM traffic per node: its hidden
// row once, its output row read + written once; the entry table (12-100 k entries x 20 B) and the element's weight block stay in L2.
// Hand-written HIP for gfx950 (CDNA4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

#define C3_ENT_I32 5             // {x, i, j, kappa (global over the targets), value (float bits)}

__global__ __launch_bounds__(256) void sym_contraction3_kernel(const float* __restrict__ h, int64_t hs, const int64_t* __restrict__ z, int C, int num_ell,
                                                               const int* __restrict__ ell_off, int nout, const int* __restrict__ out_off,
                                                               const int* __restrict__ ptr3, const int* __restrict__ ent3,
                                                               const float* __restrict__ W3, int K3, float* __restrict__ out, int64_t os) {
    extern __shared__ float xs[];                              // [num_ell][C]
    const int64_t b = blockIdx.x;
    const float* __restrict__ hb = h + b * hs;
    for (int i = threadIdx.x; i < num_ell * C; i += blockDim.x) {
        const int ell = i / C, c = i - ell * C;
        xs[i] = hb[ell_off[ell] + c];
    }
    __syncthreads();
    const float* __restrict__ w3 = W3 + z[b] * (int64_t)K3 * C;
    for (int idx = threadIdx.x; idx < nout * C; idx += blockDim.x) {
        const int o = idx / C, c = idx - o * C;
        float acc = 0.f;
        for (int e = ptr3[o]; e < ptr3[o + 1]; ++e) {
            const int* __restrict__ t = ent3 + (int64_t)e * C3_ENT_I32;
            const float w = __int_as_float(t[4]) * w3[t[3] * C + c];
            acc = fmaf(w * xs[t[0] * C + c] * xs[t[1] * C + c], xs[t[2] * C + c], acc);
        }
        out[b * os + out_off[o] + c] += acc;
    }
}

extern "C" int hg_sym_contraction3(const float* h, int64_t h_stride, const int64_t* z, int64_t N, int C, int num_ell, const int32_t* ell_off,
                                   int nout, const int32_t* out_off, const int32_t* ptr3, const int32_t* ent3, const float* W3, int K3,
                                   float* out, int64_t out_stride, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (N <= 0) return 0;
    if (!h || !z || !ell_off || !out_off || !ptr3 || !ent3 || !W3 || !out) return hg_fail(-1, "hg_sym_contraction3: null pointer");
    const size_t lds = (size_t)num_ell * C * sizeof(float);
    if (lds > 64 * 1024) return hg_fail(-2, "hg_sym_contraction3: hidden features too wide for the LDS-resident kernel");
    sym_contraction3_kernel<<<dim3((unsigned)N), 256, lds, (hipStream_t)stream>>>(h, h_stride, z, C, num_ell, ell_off, nout, out_off, ptr3, ent3, W3, K3,
                                                                                  out, out_stride);
    return hg_check_launch("hg_sym_contraction3");
}

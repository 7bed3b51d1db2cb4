// head.hip -- pair read-out: ham-irrep coefficients -> orbital blocks (gfx950).  HBM-bound elementwise/gather work:
// coalesced row reads, CSR-sparse Clebsch-Gordan expansion staged through LDS, one pass for symmetrise + H0 + mask.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

// stage 1: un-rotate every ham irrep back to the global frame (edges only), expand with sqrt(2L+1) w3j(li,lj,L)
// (hamgnn_output.py:851-891) and apply the DFT-code orbital permutation / sign flips (:1056-1096) -- the latter two are
// folded into the CSR table on the host (hamgnn_amd/plan.py:ham_tables).
__global__ __launch_bounds__(256) void ham_merge_kernel(const float* __restrict__ coeff, int64_t cs, const float* __restrict__ wig,
                                                        int nW, const HgWigOff wo, const int4* __restrict__ slot_tab,
                                                        const int* __restrict__ cg_ptr, const int* __restrict__ cg_idx,
                                                        const float* __restrict__ cg_val, int nao2, float* __restrict__ Hraw) {
    extern __shared__ float coef[];
    const int64_t e = blockIdx.x;
    const float* __restrict__ c = coeff + e * cs;
    const float* __restrict__ D = wig ? wig + e * nW : nullptr;
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
        const int4 t = slot_tab[q];                  // {L, a, base (planar index of component 0), component stride}
        float acc;
        if (D) {
            const int n = 2 * t.x + 1;
            const float* __restrict__ Dl = D + wo.o[t.x];
            acc = 0.f;
            for (int m = 0; m < n; ++m) acc = fmaf(Dl[m * n + t.y], c[t.z + m * t.w], acc);
        } else {
            acc = c[t.z + t.y * t.w];
        }
        coef[q] = acc;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
        float acc = 0.f;
        for (int k = cg_ptr[q]; k < cg_ptr[q + 1]; ++k) acc = fmaf(cg_val[k], coef[cg_idx[k]], acc);
        Hraw[e * nao2 + q] = acc;
    }
}

extern "C" int hg_ham_merge(const float* coeff, int64_t c_stride, const float* wig, int nW, const int32_t* wig_off,
                            const int32_t* slot_tab, int nslots, const int32_t* cg_ptr, const int32_t* cg_idx, const float* cg_val,
                            int nao2, int64_t rows, float* Hraw, void* stream) {
    if (rows <= 0) return 0;
    if (nslots != nao2) return hg_fail(-2, "hg_ham_merge: slot table must have nao^2 entries");
    HgWigOff wo;
    for (int i = 0; i < 8; ++i) wo.o[i] = wig_off ? wig_off[i] : 0;
    ham_merge_kernel<<<dim3((unsigned)rows), 256, sizeof(float) * (size_t)nao2, (hipStream_t)stream>>>(
        coeff, c_stride, wig, nW, wo, (const int4*)slot_tab, cg_ptr, cg_idx, cg_val, nao2, Hraw);
    return hg_check_launch("hg_ham_merge");
}

// stage 2: H = mask * (0.5 (Hraw[e] + sign Hraw[inv e]^T) + H0)   (hamgnn_output.py:1231-1285, 3782-3795, 2288-2365)
__global__ __launch_bounds__(256) void ham_finish_kernel(const float* __restrict__ Hraw, const int64_t* __restrict__ inv,
                                                         const float* __restrict__ H0, const float* __restrict__ orb_mask,
                                                         const int64_t* __restrict__ z, const int64_t* __restrict__ ia,
                                                         const int64_t* __restrict__ ib, int nao, float sign, int symmetrize,
                                                         float* __restrict__ H) {
    const int64_t e = blockIdx.x;
    const int nao2 = nao * nao;
    const int64_t eo = inv ? inv[e] : e;
    const float* __restrict__ ma = orb_mask ? orb_mask + z[ia ? ia[e] : e] * nao : nullptr;
    const float* __restrict__ mb = orb_mask ? orb_mask + z[ib ? ib[e] : e] * nao : nullptr;
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
        const int r = q / nao, c = q - r * nao;
        float v = Hraw[e * nao2 + q];
        if (symmetrize) v = 0.5f * (v + sign * Hraw[eo * nao2 + c * nao + r]);
        if (H0) v += H0[e * nao2 + q];
        if (ma) v *= ma[r] * mb[c];
        H[e * nao2 + q] = v;
    }
}

extern "C" int hg_ham_finish(const float* Hraw, const int64_t* inv, const float* H0, const float* orb_mask, const int64_t* z,
                             const int64_t* idx_a, const int64_t* idx_b, int nao, float sign, int symmetrize, int64_t rows, float* H,
                             void* stream) {
    if (rows <= 0) return 0;
    ham_finish_kernel<<<dim3((unsigned)rows), 256, 0, (hipStream_t)stream>>>(Hraw, inv, H0, orb_mask, z, idx_a, idx_b, nao, sign, symmetrize, H);
    return hg_check_launch("hg_ham_finish");
}

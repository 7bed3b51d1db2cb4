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
                                                        const float* __restrict__ cg_val, int nslots, int nout,
                                                        float* __restrict__ Hraw) {
    extern __shared__ float coef[];
    const int64_t e = blockIdx.x;
    const float* __restrict__ c = coeff + e * cs;
    const float* __restrict__ D = wig ? wig + e * nW : nullptr;
    for (int q = threadIdx.x; q < nslots; q += blockDim.x) {
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
    for (int q = threadIdx.x; q < nout; q += blockDim.x) {
        float acc = 0.f;
        for (int k = cg_ptr[q]; k < cg_ptr[q + 1]; ++k) acc = fmaf(cg_val[k], coef[cg_idx[k]], acc);
        Hraw[e * nout + q] = acc;
    }
}

extern "C" int hg_ham_merge(const float* coeff, int64_t c_stride, const float* wig, int nW, const int32_t* wig_off,
                            const int32_t* slot_tab, int nslots, const int32_t* cg_ptr, const int32_t* cg_idx, const float* cg_val,
                            int nout, int64_t rows, float* Hraw, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nslots <= 0 || nout <= 0 || nslots > 16384) return hg_fail(-2, "hg_ham_merge: bad slot / output count");
    HgWigOff wo;
    for (int i = 0; i < 8; ++i) wo.o[i] = wig_off ? wig_off[i] : 0;
    ham_merge_kernel<<<dim3((unsigned)rows), 256, sizeof(float) * (size_t)nslots, (hipStream_t)stream>>>(
        coeff, c_stride, wig, nW, wo, (const int4*)slot_tab, cg_ptr, cg_idx, cg_val, nslots, nout, Hraw);
    return hg_check_launch("hg_ham_merge");
}

// stage 2: H = mask * (0.5 (Hraw[e] + sign Hraw[inv e]^T) + H0)   (hamgnn_output.py:1231-1285, 3782-3795, 2288-2365);
// flags bit 1: H0 is added after the mask (SOC branches, :3603-3609).  The orbital-mask row index wraps at mask_w so that the
// same kernel finishes the (2 nao)^2 spin-block matrices of the su2 branch (:3163-3168) with sign = +1 (real) / -1 (imag).
__global__ __launch_bounds__(256) void ham_finish_kernel(const float* __restrict__ Hraw, int64_t hs, const int64_t* __restrict__ inv,
                                                         const float* __restrict__ H0, const float* __restrict__ orb_mask,
                                                         const int64_t* __restrict__ z, const int64_t* __restrict__ ia,
                                                         const int64_t* __restrict__ ib, int nao, int mask_w, float sign, int flags,
                                                         float* __restrict__ H) {
    const int64_t e = blockIdx.x;
    const int nao2 = nao * nao;
    const int64_t eo = inv ? inv[e] : e;
    const float* __restrict__ ma = orb_mask ? orb_mask + z[ia ? ia[e] : e] * mask_w : nullptr;
    const float* __restrict__ mb = orb_mask ? orb_mask + z[ib ? ib[e] : e] * mask_w : nullptr;
    const bool sym = flags & 1, h0_last = flags & 2;
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
        const int r = q / nao, c = q - r * nao;
        float v = Hraw[e * hs + q];
        if (sym) v = 0.5f * (v + sign * Hraw[eo * hs + c * nao + r]);
        if (H0 && !h0_last) v += H0[e * nao2 + q];
        if (ma) v *= ma[r % mask_w] * mb[c % mask_w];
        if (H0 && h0_last) v += H0[e * nao2 + q];
        H[e * nao2 + q] = v;
    }
}

extern "C" int hg_ham_finish(const float* Hraw, int64_t h_stride, const int64_t* inv, const float* H0, const float* orb_mask,
                             int mask_w, const int64_t* z, const int64_t* idx_a, const int64_t* idx_b, int nao, float sign, int flags,
                             int64_t rows, float* H, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (orb_mask && (mask_w <= 0 || nao % mask_w)) return hg_fail(-2, "hg_ham_finish: nao must be a multiple of the mask width");
    ham_finish_kernel<<<dim3((unsigned)rows), 256, 0, (hipStream_t)stream>>>(Hraw, h_stride, inv, H0, orb_mask, z, idx_a, idx_b, nao,
                                                                             mask_w > 0 ? mask_w : nao, sign, flags, H);
    return hg_check_launch("hg_ham_finish");
}

// ------------------------------------------------------------------------------------------------ one-pass read-out (non-SOC head)
// stage 1 + stage 2 in ONE pass over the rows.  A block walks groups of HR_PAIRS (edge, inverse edge) PAIRS: the rows' coefficient and
// Wigner rows are pulled into the LDS with coalesced loads (all rows of the group in flight together), un-rotated and CG-expanded
// LDS -> LDS, the two merged blocks of a pair are symmetrised against each other, + H0, masked, and written straight into the caller's
// [rows, nao^2] result.  Against hg_ham_merge + hg_ham_finish this removes the Hraw round trip (1 write + 2 reads of nao^2 per row),
// one launch, the dependent per-element gathers and the per-row re-reading of the CSR table (staged once per block).  On-site rows pair
// with themselves.
#ifndef HR_PAIRS
#define HR_PAIRS 2                // measured r3: 4 pairs per step (LDS per block doubles, fewer resident blocks) 3.4 -> 6.1 ms per 822 k rows
#endif
__global__ __launch_bounds__(256) void ham_readout_kernel(const float* __restrict__ coeff, int64_t cs, int cw, const float* __restrict__ wig,
                                                          int nW, int nWuse, const HgWigOff wo, const int4* __restrict__ slot_tab,
                                                          const int* __restrict__ cg_ptr, const int* __restrict__ cg_idx,
                                                          const float* __restrict__ cg_val, int nslots, int nao, int nnz,
                                                          const int64_t* __restrict__ pair_a, const int64_t* __restrict__ pair_b,
                                                          int64_t npairs, const float* __restrict__ H0, const float* __restrict__ orb_mask,
                                                          int mask_w, const int64_t* __restrict__ z, const int64_t* __restrict__ ia,
                                                          const int64_t* __restrict__ ib, float sign, int flags, float* __restrict__ H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int R = 2 * HR_PAIRS;                            // rows per step
    const int nao2 = nao * nao;
    const int rawlen = cw + nWuse;                             // per row: coefficient row, then the Wigner blocks l <= lmax(ham irreps)
    const int rlen = rawlen > nao2 ? rawlen : nao2;            // the merged block re-uses the raw buffer
    int* __restrict__ s_ptr = reinterpret_cast<int*>(sm);
    int* __restrict__ s_idx = s_ptr + nao2 + 1;
    float* __restrict__ s_val = reinterpret_cast<float*>(s_idx + nnz);
    int* __restrict__ s_rc = reinterpret_cast<int*>(s_val + nnz);      // [nao2] (row | col << 8 | transposed index << 16) of every element
    float* __restrict__ s_raw = reinterpret_cast<float*>(s_rc + nao2); // [R][rlen]   raw rows, later the merged nao x nao blocks
    float* __restrict__ s_coef = s_raw + R * rlen;             // [R][nslots] un-rotated coefficients
    float* __restrict__ s_mask = s_coef + R * nslots;          // [R][2][nao] orbital masks of the row's two atoms
    for (int i = threadIdx.x; i <= nao2; i += blockDim.x) s_ptr[i] = cg_ptr[i];
    for (int i = threadIdx.x; i < nnz; i += blockDim.x) {
        s_idx[i] = cg_idx[i];
        s_val[i] = cg_val[i];
    }
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
        const int rr = q / nao, cc = q - rr * nao;
        s_rc[q] = rr | (cc << 8) | ((cc * nao + rr) << 16);
    }
    const bool sym = flags & 1, h0_last = flags & 2;
    const int64_t ngroups = (npairs + HR_PAIRS - 1) / HR_PAIRS;
    // Software pipeline over the block's groups (r3, late): the rows of group i + 1 -- coefficient row, Wigner row, masks -- and the H0
    // values of group i are requested into registers at the top of group i and land under its three LDS -> LDS stages; before, every
    // group exposed its row-index load, its row loads and its H0 loads one after the other (3.4 ms per 822 k rows for 4 GB of traffic).
    // Register slots: HR_CJ x 256 coefficient floats, HR_WJ x 256 Wigner floats, HR_HJ x 256 H0 values per row (wider rows: direct loads).
    constexpr int HR_CJ = 2, HR_WJ = 1, HR_HJ = 2;
    float pc[R][HR_CJ], pw[R][HR_WJ], pm[R];
    int64_t rown[R];                                           // the next group's rows (uniform)
    int zn[R][2];                                              // ... and the species of their two atoms (uniform: scalar loads -- as a
                                                               // per-thread chain atom -> species -> mask it put a vmcnt(0) into the requests)
    auto rows_of = [&](int64_t gq, int64_t (&rw)[R]) {       // three rounds of independent scalar loads (pairs -> atoms -> species), no branch
        int64_t ra[HR_PAIRS], rb[HR_PAIRS];                    // between them: invalid slots read row 0 and are dropped afterwards
        bool ok[HR_PAIRS];
#pragma unroll
        for (int pr = 0; pr < HR_PAIRS; ++pr) {
            const int64_t p = gq * HR_PAIRS + pr;
            ok[pr] = gq < ngroups && p < npairs;
            const int64_t pcl = ok[pr] ? p : 0;
            ra[pr] = pair_a ? pair_a[pcl] : pcl;
            rb[pr] = pair_b ? pair_b[pcl] : ra[pr];
        }
        int64_t aa[R], bb[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = (r & 1) ? rb[r >> 1] : ra[r >> 1];
            rw[r] = !ok[r >> 1] ? -1 : ((r & 1) && rb[r >> 1] == ra[r >> 1]) ? -1 : row;      // self-paired rows occupy slot 0 of their pair only
            aa[r] = ia ? ia[row] : row;
            bb[r] = ib ? ib[row] : row;
        }
        if (orb_mask) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                zn[r][0] = (int)z[aa[r]];
                zn[r][1] = (int)z[bb[r]];
            }
        }
    };
    auto request = [&](const int64_t (&rw)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t e = rw[r];
            if (e < 0) continue;
            const float* __restrict__ c = coeff + e * cs;
#pragma unroll
            for (int j = 0; j < HR_CJ; ++j) {
                const int k = threadIdx.x + j * 256;
                if (k < cw) pc[r][j] = c[k];
            }
            if (wig) {
                const float* __restrict__ D = wig + e * nW;
#pragma unroll
                for (int j = 0; j < HR_WJ; ++j) {
                    const int k = threadIdx.x + j * 256;
                    if (k < nWuse) pw[r][j] = D[k];
                }
            }
            if (orb_mask && threadIdx.x < 2 * nao) {
                const int side = threadIdx.x >= nao, o = threadIdx.x - side * nao;
                pm[r] = orb_mask[(side ? zn[r][1] : zn[r][0]) * mask_w + o % mask_w];
            }
        }
    };
    rows_of(blockIdx.x, rown);
    request(rown);
    for (int64_t gp = blockIdx.x; gp < ngroups; gp += gridDim.x) {
        int64_t rowc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rowc[r] = rown[r];
        __syncthreads();                                       // tables staged / previous group's buffers free
        // ---- the group's rows: registers -> LDS (wider rows than the register slots: the rest straight from memory)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t e = rowc[r];
            float* __restrict__ dst = s_raw + r * rlen;
            if (e < 0) continue;
#pragma unroll
            for (int j = 0; j < HR_CJ; ++j) {
                const int k = threadIdx.x + j * 256;
                if (k < cw) dst[k] = pc[r][j];
            }
            const float* __restrict__ c = coeff + e * cs;
            for (int k = threadIdx.x + HR_CJ * 256; k < cw; k += blockDim.x) dst[k] = c[k];
            if (wig) {
#pragma unroll
                for (int j = 0; j < HR_WJ; ++j) {
                    const int k = threadIdx.x + j * 256;
                    if (k < nWuse) dst[cw + k] = pw[r][j];
                }
                const float* __restrict__ D = wig + e * nW;
                for (int k = threadIdx.x + HR_WJ * 256; k < nWuse; k += blockDim.x) dst[cw + k] = D[k];
            }
            if (orb_mask && threadIdx.x < 2 * nao) {
                const int side = threadIdx.x >= nao, o = threadIdx.x - side * nao;
                s_mask[(r * 2 + side) * nao + o] = pm[r];
            }
        }
        // ---- requests that travel under this group's stages: its H0 values, the next group's rows
        float ph[R][HR_HJ];
        if (H0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (rowc[r] < 0) continue;
#pragma unroll
                for (int j = 0; j < HR_HJ; ++j) {
                    const int q = threadIdx.x + j * 256;
                    if (q < nao2) ph[r][j] = H0[rowc[r] * nao2 + q];
                }
            }
        }
        rows_of(gp + gridDim.x, rown);
        request(rown);
        __syncthreads();
        // ---- un-rotate: coef[q] = sum_m D^L[m][a] c[L slot][m]
        for (int q = threadIdx.x; q < nslots; q += blockDim.x) {
            const int4 t = slot_tab[q];                        // {L, a, base (planar index of component 0), component stride}
            const int n = 2 * t.x + 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (rowc[r] < 0) continue;
                const float* __restrict__ c = s_raw + r * rlen;
                float acc;
                if (wig) {
                    const float* __restrict__ Dl = c + cw + wo.o[t.x];
                    acc = 0.f;                                 // (compile-time trip counts per L through a switch: the slots of a wave span several
                    for (int m = 0; m < n; ++m) acc = fmaf(Dl[m * n + t.y], c[t.z + m * t.w], acc);       //  L, the divergent switch ran 4.3 vs 2.45 ms)
                } else {
                    acc = c[t.z + t.y * t.w];
                }
                s_coef[r * nslots + q] = acc;
            }
        }
        __syncthreads();
        // ---- CG expansion (+ reorder, signs: folded into the CSR table) into the raw buffer: one table walk serves all rows
        for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0.f;
            for (int k = s_ptr[q]; k < s_ptr[q + 1]; ++k) {
                const float v = s_val[k];
                const int j = s_idx[k];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fmaf(v, s_coef[r * nslots + j], acc[r]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) s_raw[r * rlen + q] = acc[r];
        }
        __syncthreads();
        // ---- symmetrise against the partner row, + H0, mask, store
        auto finish = [&](int q, int r, float h0) {
            const int64_t e = rowc[r];
            const int code = s_rc[q];
            const int rr = code & 0xff, cc = (code >> 8) & 0xff, qt = code >> 16;
            const int rp = rowc[r ^ 1] >= 0 ? (r ^ 1) : r;     // partner slot (itself for self-paired rows)
            float v = s_raw[r * rlen + q];
            if (sym) v = 0.5f * (v + sign * s_raw[rp * rlen + qt]);
            if (!h0_last) v += h0;
            if (orb_mask) v *= s_mask[(r * 2) * nao + rr] * s_mask[(r * 2 + 1) * nao + cc];
            if (h0_last) v += h0;
            H[e * nao2 + q] = v;
        };
#pragma unroll
        for (int j = 0; j < HR_HJ; ++j) {
            const int q = threadIdx.x + j * 256;
            if (q < nao2) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (rowc[r] >= 0) finish(q, r, H0 ? ph[r][j] : 0.f);
            }
        }
        for (int q = threadIdx.x + HR_HJ * 256; q < nao2; q += blockDim.x) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (rowc[r] >= 0) finish(q, r, H0 ? H0[rowc[r] * nao2 + q] : 0.f);
        }
    }
}

extern "C" int hg_ham_readout(const float* coeff, int64_t c_stride, int c_width, const float* wig, int nW, const int32_t* wig_off,
                              int lmax_ham, const int32_t* slot_tab, int nslots, const int32_t* cg_ptr, const int32_t* cg_idx,
                              const float* cg_val, int nnz, int nao, const int64_t* pair_a, const int64_t* pair_b, int64_t npairs,
                              const float* H0, const float* orb_mask, int mask_w, const int64_t* z, const int64_t* idx_a, const int64_t* idx_b,
                              float sign, int flags, float* H, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (npairs <= 0) return 0;
    const int nao2 = nao * nao;
    if (nslots <= 0 || nao <= 0 || nnz <= 0 || c_width <= 0) return hg_fail(-2, "hg_ham_readout: bad table sizes");
    if (orb_mask && (mask_w <= 0 || nao % mask_w)) return hg_fail(-2, "hg_ham_readout: nao must be a multiple of the mask width");
    if (lmax_ham < 0 || lmax_ham > 7) return hg_fail(-2, "hg_ham_readout: lmax of the hamiltonian irreps must be 0..7");
    HgWigOff wo;
    int nWuse = 0;
    for (int i = 0; i < 8; ++i) wo.o[i] = wig_off ? wig_off[i] : 0;
    if (wig) {
        nWuse = wo.o[lmax_ham] + (2 * lmax_ham + 1) * (2 * lmax_ham + 1);
        if (nWuse > nW) return hg_fail(-2, "hg_ham_readout: Wigner rows do not reach lmax of the hamiltonian irreps");
    }
    const int rawlen = c_width + nWuse, rlen = rawlen > nao2 ? rawlen : nao2;
    if (nao > 181) return hg_fail(-2, "hg_ham_readout: nao too large for the packed index table");
    const size_t lds = sizeof(float) * ((size_t)(2 * nao2 + 1) + 2 * (size_t)nnz + (size_t)(2 * HR_PAIRS) * ((size_t)rlen + (size_t)nslots + 2 * (size_t)nao));
    if (lds > 64 * 1024) return hg_fail(-2, "hg_ham_readout: tables exceed 64 KB of LDS (use hg_ham_merge + hg_ham_finish)");
    const int64_t ngroups = (npairs + HR_PAIRS - 1) / HR_PAIRS;
    const int64_t blocks = ngroups < 256 * 5 ? ngroups : 256 * 5;    // persistent blocks (~5 per CU fit the LDS), each walks groups of pairs
    ham_readout_kernel<<<dim3((unsigned)blocks), 256, lds, (hipStream_t)stream>>>(
        coeff, c_stride, c_width, wig, nW, nWuse, wo, (const int4*)slot_tab, cg_ptr, cg_idx, cg_val, nslots, nao, nnz, pair_a, pair_b, npairs, H0,
        orb_mask, mask_w > 0 ? mask_w : nao, z, idx_a, idx_b, sign, flags, H);
    return hg_check_launch("hg_ham_readout");
}

// ------------------------------------------------------------------------------------------------ SOC / so3 (a18)
// symmetrize_orbital_coefficients (hamgnn_output.py:2367-2431): every element -> mean over its (row shell, col shell) block.
// tab: int32[nao2][4] = {r0, r1, c0, c1} of the block the element belongs to.
__global__ __launch_bounds__(256) void block_mean_kernel(const float* __restrict__ x, int64_t xs, const int4* __restrict__ tab, int nao,
                                                         float* __restrict__ out) {
    extern __shared__ float sm[];
    const int64_t e = blockIdx.x;
    const int nao2 = nao * nao;
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) sm[q] = x[e * xs + q];
    __syncthreads();
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
        const int4 t = tab[q];
        float acc = 0.f;
        for (int r = t.x; r < t.y; ++r)
            for (int c = t.z; c < t.w; ++c) acc += sm[r * nao + c];
        out[e * nao2 + q] = acc / (float)((t.y - t.x) * (t.w - t.z));
    }
}

extern "C" int hg_block_mean(const float* x, int64_t x_stride, const int32_t* tab, int nao, int64_t rows, float* out, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    block_mean_kernel<<<dim3((unsigned)rows), 256, sizeof(float) * (size_t)nao * nao, (hipStream_t)stream>>>(x, x_stride, (const int4*)tab, nao, out);
    return hg_check_launch("hg_block_mean");
}

// spin-block assembly of the so3 SOC Hamiltonian (hamgnn_output.py:3076-3144, 3603-3609):
//   A_k = antiherm(ksi * L[..., k]) = 0.5 (ksi L_k - (ksi L_k)[inv]^T)          (k: 0 = x, 1 = y, 2 = z)
//   real = [[H, A_1], [A_1, H]] + H0r ;  imag = [[A_2, A_0], [-A_0, -A_2]] + H0i
// zero_diag != 0 (add_H_nonsoc): the spin-diagonal blocks of H0r are not added (:3034-3049).
__global__ __launch_bounds__(256) void soc_assemble_kernel(const float* __restrict__ H, const float* __restrict__ ksi, const float* __restrict__ L,
                                                           const int64_t* __restrict__ inv, const float* __restrict__ H0r,
                                                           const float* __restrict__ H0i, int nao, int symmetrize, int zero_diag,
                                                           float* __restrict__ outr, float* __restrict__ outi) {
    const int64_t e = blockIdx.x;
    const int64_t eo = inv ? inv[e] : e;
    const int nao2 = nao * nao, n2 = 2 * nao;
    const int64_t big = (int64_t)n2 * n2;
    for (int q = threadIdx.x; q < (int)big; q += blockDim.x) {
        const int R = q / n2, Cc = q - R * n2;
        const int sr = R >= nao, sc = Cc >= nao;
        const int r = R - sr * nao, c = Cc - sc * nao;
        const int el = r * nao + c, elT = c * nao + r;
        const int k = (sr == sc) ? 2 : 0;                      // imag: diagonal blocks use L_z, off-diagonal L_x
        const float kv = ksi[e * nao2 + el], kvT = ksi[eo * nao2 + elT];
        float ai = kv * L[(e * nao2 + el) * 3 + k];
        float ar = kv * L[(e * nao2 + el) * 3 + 1];
        if (symmetrize) {
            ai = 0.5f * (ai - kvT * L[(eo * nao2 + elT) * 3 + k]);
            ar = 0.5f * (ar - kvT * L[(eo * nao2 + elT) * 3 + 1]);
        }
        float vr = (sr == sc) ? H[e * nao2 + el] : ar;
        float vi = (sr == sc) ? (sr ? -ai : ai) : (sr ? -ai : ai);   // (0,0): +A2, (1,1): -A2, (0,1): +A0, (1,0): -A0
        if (H0r && !(zero_diag && sr == sc)) vr += H0r[e * big + q];
        if (H0i) vi += H0i[e * big + q];
        outr[e * big + q] = vr;
        outi[e * big + q] = vi;
    }
}

extern "C" int hg_soc_assemble(const float* H, const float* ksi, const float* L, const int64_t* inv, const float* H0r, const float* H0i,
                               int nao, int symmetrize, int zero_diag, int64_t rows, float* out_real, float* out_imag, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    soc_assemble_kernel<<<dim3((unsigned)rows), 256, 0, (hipStream_t)stream>>>(H, ksi, L, inv, H0r, H0i, nao, symmetrize, zero_diag, out_real, out_imag);
    return hg_check_launch("hg_soc_assemble");
}

// ------------------------------------------------------------------------------------------------ zero-point shift (a17)
// hamgnn_output.py:3971-3981 (non-SOC) / :3892-3913 (SOC, spin-diagonal real blocks):
//   dE = sum_{S > thr} (H - Href) / sum_{S > thr} S ;  H -= dE * S          (one dE per batch, fp64 accumulation)
// SOC (soc != 0): H, Href are [(2 nao)^2] rows, S is [nao^2]; the difference is (uu + dd) - (uu_ref + dd_ref), the
// denominator 2 sum S, and both spin-diagonal blocks are shifted.
__device__ __forceinline__ int64_t zp_index(int64_t i, int nao, int soc, int blk) {
    if (!soc) return i;
    const int64_t n2 = (int64_t)nao * nao, row = i / n2;
    const int q = (int)(i - row * n2), a = q / nao, b = q - a * nao;
    return row * 4 * n2 + (int64_t)(blk * nao + a) * (2 * nao) + blk * nao + b;
}

__global__ __launch_bounds__(256) void zp_partial_kernel(const float* __restrict__ H, const float* __restrict__ Href,
                                                         const float* __restrict__ S, int64_t count, int nao, int soc, float thr,
                                                         double* __restrict__ partial) {
    __shared__ double sn[256], sd[256];
    double num = 0.0, den = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = S[i];
        if (s > thr) {
            const int64_t i0 = zp_index(i, nao, soc, 0);
            double d = (double)H[i0] - (double)Href[i0];
            if (soc) {
                const int64_t i1 = zp_index(i, nao, soc, 1);
                d += (double)H[i1] - (double)Href[i1];
            }
            num += d;
            den += (double)s;
        }
    }
    sn[threadIdx.x] = num; sd[threadIdx.x] = den;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { sn[threadIdx.x] += sn[threadIdx.x + w]; sd[threadIdx.x] += sd[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sn[0]; partial[2 * blockIdx.x + 1] = sd[0]; }
}

__global__ __launch_bounds__(256) void zp_apply_kernel(float* __restrict__ H, const float* __restrict__ S, int64_t count, int nao, int soc,
                                                       const double* __restrict__ partial, int nparts, float* __restrict__ shift_out) {
    __shared__ double sh_shift;
    if (threadIdx.x == 0) {                                     // fixed-order reduction of the block partials: deterministic
        double num = 0.0, den = 0.0;
        for (int p = 0; p < nparts; ++p) { num += partial[2 * p]; den += partial[2 * p + 1]; }
        sh_shift = num / (soc ? 2.0 * den : den);
        if (blockIdx.x == 0 && shift_out) *shift_out = (float)sh_shift;
    }
    __syncthreads();
    const float dE = (float)sh_shift;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = S[i];
        const int64_t i0 = zp_index(i, nao, soc, 0);
        H[i0] -= dE * s;
        if (soc) { const int64_t i1 = zp_index(i, nao, soc, 1); H[i1] -= dE * s; }
    }
}

extern "C" int hg_zero_point_shift(float* H, const float* Href, const float* S, int64_t rows, int nao, int soc, float threshold,
                                   double* partial_scratch, int nparts, float* shift_out, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nparts <= 0 || nparts > 1024) return hg_fail(-2, "hg_zero_point_shift: scratch must hold 1..1024 partial pairs");
    const int64_t count = rows * nao * nao;
    zp_partial_kernel<<<dim3((unsigned)nparts), 256, 0, (hipStream_t)stream>>>(H, Href, S, count, nao, soc, threshold, partial_scratch);
    const unsigned grid = (unsigned)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    zp_apply_kernel<<<dim3(grid), 256, 0, (hipStream_t)stream>>>(H, S, count, nao, soc, partial_scratch, nparts, shift_out);
    return hg_check_launch("hg_zero_point_shift");
}

// ------------------------------------------------------------------------------------------------ correlation product (a21)
// MACE symmetric contraction with correlation <= 2 on the nodes (hamgnn/nn/interaction_blocks.py:234-260 ->
// toolbox/mace/modules/symmetric_contraction.py:212-230), sparse form of the reference's dense einsums:
//   out[o, c] = sum_x ( sum_(e in row1(o)) U1 W1[z, kap, c]  +  sum_(e in row2(o)) U2 W2[z, kap, c] x[c, i] ) x[c, x]
// One workgroup per node; the node's hidden features x[c][ell] sit in LDS; one thread per (output element o, channel c).
// Node-level and off in the shipped configurations: written for correctness and coalesced weight reads, not tuned.
__global__ __launch_bounds__(256) void sym_contraction_kernel(const float* __restrict__ h, int64_t hs, const int64_t* __restrict__ z, int C, int num_ell,
                                                              const int* __restrict__ ell_off, int nout, const int* __restrict__ out_off,
                                                              const int* __restrict__ ptr1, const int4* __restrict__ ent1,
                                                              const int* __restrict__ ptr2, const int4* __restrict__ ent2,
                                                              const float* __restrict__ W1, int K1, const float* __restrict__ W2, int K2,
                                                              float* __restrict__ out, int64_t os) {
    extern __shared__ float xs[];                              // [num_ell][C]
    const int64_t b = blockIdx.x;
    const float* __restrict__ hb = h + b * hs;
    for (int i = threadIdx.x; i < num_ell * C; i += blockDim.x) {
        const int ell = i / C, c = i - ell * C;
        xs[i] = hb[ell_off[ell] + c];
    }
    __syncthreads();
    const int64_t zb = z[b];
    const float* __restrict__ w1 = W1 + zb * (int64_t)K1 * C;
    const float* __restrict__ w2 = W2 + zb * (int64_t)K2 * C;
    for (int idx = threadIdx.x; idx < nout * C; idx += blockDim.x) {
        const int o = idx / C, c = idx - o * C;
        float acc = 0.f;
        for (int e = ptr1[o]; e < ptr1[o + 1]; ++e) {
            const int4 t = ent1[e];                            // {x, kappa, -, value}
            acc = fmaf(__int_as_float(t.w) * w1[t.y * C + c], xs[t.x * C + c], acc);
        }
        for (int e = ptr2[o]; e < ptr2[o + 1]; ++e) {
            const int4 t = ent2[e];                            // {x, i, kappa, value}
            acc = fmaf(__int_as_float(t.w) * w2[t.z * C + c] * xs[t.y * C + c], xs[t.x * C + c], acc);
        }
        out[b * os + out_off[o] + c] = acc;
    }
}

extern "C" int hg_sym_contraction(const float* h, int64_t h_stride, const int64_t* z, int64_t N, int C, int num_ell, const int32_t* ell_off,
                                  int nout, const int32_t* out_off, const int32_t* ptr1, const int32_t* ent1, const int32_t* ptr2,
                                  const int32_t* ent2, const float* W1, int K1, const float* W2, int K2, float* out, int64_t out_stride,
                                  void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (N <= 0) return 0;
    const size_t lds = sizeof(float) * (size_t)num_ell * C;
    if (lds > 64 * 1024) return hg_fail(-2, "hg_sym_contraction: hidden features too wide for the LDS-resident kernel");
    sym_contraction_kernel<<<dim3((unsigned)N), 256, lds, (hipStream_t)stream>>>(h, h_stride, z, C, num_ell, ell_off, nout, out_off, ptr1,
                                                                                 (const int4*)ent1, ptr2, (const int4*)ent2, W1, K1, W2, K2, out,
                                                                                 out_stride);
    return hg_check_launch("hg_sym_contraction");
}

// ------------------------------------------------------------------------------------------------ k-space assembly (f4: band energies)
// H(k)[i a, j b] = delta_ij H_on[i][a][b] + sum_{e: i -> j} exp(2 pi i k . nbr_shift_e) H_off[e][a][b]  of ONE crystal, written straight in
// the COMPACT orbital basis (orbitals an element does not have are skipped: the reference builds the nao_max-padded matrix and
// masked_selects it, hamgnn_output.py:1776-1905).  The reference accumulates with index_put(accumulate=True) (atomics); here the edges
// are grouped by atom pair on the host (index plumbing) and one block owns one (pair, k): fixed summation order, no atomics.
//   pair_ptr[npairs+1], pair_edges[]: edges of every (i, j) pair (crystal-local edge ids);  pair_ij[npairs][2];
//   orank[n][nao]: rank of orbital a inside atom i's valid set or -1;  ooff[n]: first compact index of atom i;  M: compact dimension;
//   Hk: [nk][M][M] complex64 (float2), zero-initialised by the caller.
__global__ __launch_bounds__(256) void hk_onsite_kernel(const float* __restrict__ on, int nao, const int* __restrict__ orank,
                                                        const int* __restrict__ ooff, int M, int nk, float2* __restrict__ Hk) {
    const int i = blockIdx.x, k = blockIdx.y;
    const int nao2 = nao * nao;
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
        const int a = q / nao, b = q - a * nao;
        const int ra = orank[i * nao + a], rb = orank[i * nao + b];
        if (ra < 0 || rb < 0) continue;
        Hk[((int64_t)k * M + ooff[i] + ra) * M + ooff[i] + rb] = make_float2(on[(int64_t)i * nao2 + q], 0.f);
    }
}

__global__ __launch_bounds__(256) void hk_pairs_kernel(const float* __restrict__ off, const float* __restrict__ shift, const float* __restrict__ kvec,
                                                       const int64_t* __restrict__ pair_ptr, const int64_t* __restrict__ pair_edges,
                                                       const int64_t* __restrict__ pair_ij, int nao, const int* __restrict__ orank,
                                                       const int* __restrict__ ooff, int M, float2* __restrict__ Hk) {
    const int64_t p = blockIdx.x;
    const int k = blockIdx.y;
    const int nao2 = nao * nao;
    const int64_t i = pair_ij[2 * p], j = pair_ij[2 * p + 1];
    const int64_t q0 = pair_ptr[p], q1 = pair_ptr[p + 1];
    const float kx = kvec[3 * k], ky = kvec[3 * k + 1], kz = kvec[3 * k + 2];
    for (int q = threadIdx.x; q < nao2; q += blockDim.x) {
        const int a = q / nao, b = q - a * nao;
        const int ra = orank[i * nao + a], rb = orank[j * nao + b];
        if (ra < 0 || rb < 0) continue;
        float re = 0.f, im = 0.f;
        for (int64_t t = q0; t < q1; ++t) {
            const int64_t e = pair_edges[t];
            // phase in double: |k . shift| reaches tens of turns for long bonds, and sincosf loses the fraction
            const double ph = 6.283185307179586 * ((double)kx * shift[3 * e] + (double)ky * shift[3 * e + 1] + (double)kz * shift[3 * e + 2]);
            double s, c;
            sincos(ph, &s, &c);
            const float h = off[e * nao2 + q];
            re = fmaf((float)c, h, re);
            im = fmaf((float)s, h, im);
        }
        float2* dst = Hk + ((int64_t)k * M + ooff[i] + ra) * M + ooff[j] + rb;
        const float2 old = *dst;                               // (i, i) pairs (self images) add onto the on-site block; one owner per element
        *dst = make_float2(old.x + re, old.y + im);
    }
}

extern "C" int hg_hk_assemble(const float* on, const float* off, const float* nbr_shift, const float* kvec, int nk, const int64_t* pair_ptr,
                              const int64_t* pair_edges, const int64_t* pair_ij, int64_t npairs, int n_atoms, int nao, const int32_t* orank,
                              const int32_t* ooff, int M, float* Hk, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (n_atoms <= 0 || nk <= 0 || M <= 0) return 0;
    if (nao <= 0 || nk > 65535) return hg_fail(-2, "hg_hk_assemble: bad sizes");
    hk_onsite_kernel<<<dim3((unsigned)n_atoms, (unsigned)nk), 256, 0, (hipStream_t)stream>>>(on, nao, orank, ooff, M, nk, (float2*)Hk);
    if (npairs > 0)
        hk_pairs_kernel<<<dim3((unsigned)npairs, (unsigned)nk), 256, 0, (hipStream_t)stream>>>(off, nbr_shift, kvec, pair_ptr, pair_edges, pair_ij, nao,
                                                                                             orank, ooff, M, (float2*)Hk);
    return hg_check_launch("hg_hk_assemble");
}

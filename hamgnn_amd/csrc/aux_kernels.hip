// aux_kernels.hip -- the memory-bound / small kernels around the fused edge kernel (gfx950).
// See include/hamgnn_hip.h for the reference op cluster each entry point replaces.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "hg_common.h"

static thread_local char g_err[512] = "";

int hg_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "unknown");
    return code;
}
int hg_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return -1;
    }
    return 0;
}
int hg_lds_attr_once(unsigned char* done, int dev, const void* kernel, int bytes) {
    if (dev < 0 || dev >= HG_MAX_DEVICES) return hg_fail(-3, "device index out of range");
    if (__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) return 0;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return hg_fail(-3, hipGetErrorString(e));
    __atomic_store_n(&done[dev], (unsigned char)1, __ATOMIC_RELEASE);
    return 0;
}
extern "C" const char* hg_last_error(void) { return g_err; }
extern "C" int hg_version(void) { return 2; }

// Workspace query ("no hidden allocation on the hot path -- workspace size is queried").  No entry point of this library
// allocates: outputs and scratch are caller-provided.  The two that need scratch beyond their documented outputs report its size here;
// every other entry point returns 0.  Host-only: callable without a GPU.
extern "C" int64_t hg_scratch_bytes(const char* entry_point, int64_t rows, int arg) {
    if (!entry_point || rows < 0) return -1;
    auto is = [&](const char* n) { const char* a = entry_point; while (*a && *a == *n) { ++a; ++n; } return *a == 0 && *n == 0; };
    if (is("hg_edge_geometry")) return rows * 4 * (int64_t)sizeof(float);              // ang_scratch [E][4]
    if (is("hg_zero_point_shift")) return 2 * (int64_t)(arg > 0 ? arg : 256) * (int64_t)sizeof(double);   // partial_scratch [2 nparts]
    if (is("hg_linear_wgrad")) return ((rows + 1023) / 1024) * (int64_t)(arg > 0 ? arg : 0) * 1024 * (int64_t)sizeof(float);   // partial [chunks][arg = nunits][16][64]
    return 0;
}

// ------------------------------------------------------------------------------------------------ edge geometry
// angles of R_e = Rx(beta) Ry(alpha) taking the e3nn-order unit vector n = (v_y, v_z, v_x)/|v| onto the pole (0,1,0);
// Bessel * cosine-cutoff radial basis evaluated in fp64 (phase n*pi*r/rc by Chebyshev recurrence) and rounded once.
__global__ void geometry_kernel(const float* __restrict__ pos, const int64_t* __restrict__ ei, const float* __restrict__ shift,
                                int64_t E, float cutoff, int R, float4* __restrict__ ang, float* __restrict__ rbf,
                                float* __restrict__ len) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t j = ei[e], i = ei[E + e];
    const double vx = (double)pos[3 * i + 0] + (double)shift[3 * e + 0] - (double)pos[3 * j + 0];
    const double vy = (double)pos[3 * i + 1] + (double)shift[3 * e + 1] - (double)pos[3 * j + 1];
    const double vz = (double)pos[3 * i + 2] + (double)shift[3 * e + 2] - (double)pos[3 * j + 2];
    const double r = sqrt(vx * vx + vy * vy + vz * vz);
    const double inv = r > 0 ? 1.0 / r : 0.0;
    const double nx = vy * inv, ny = vz * inv, nz = vx * inv;            // e3nn axis order = physical (y, z, x)
    const double rho = sqrt(nx * nx + nz * nz);
    double ca = 1.0, sa = 0.0;
    if (rho > 1e-12) { ca = nz / rho; sa = -nx / rho; }
    ang[e] = make_float4((float)ca, (float)sa, (float)ny, (float)(-rho));
    if (len) len[e] = (float)r;
    if (rbf) {
        const double th = M_PI * r / (double)cutoff;
        double s1, c1;
        sincos(th, &s1, &c1);
        const double fc = (r < (double)cutoff) ? 0.5 * (c1 + 1.0) * inv : 0.0;
        double sn = 0.0, cn = 1.0;
        for (int n = 0; n < R; ++n) {
            const double s2 = sn * c1 + cn * s1, c2 = cn * c1 - sn * s1;
            sn = s2; cn = c2;
            rbf[e * R + n] = (float)(sn * fc);
        }
    }
}

// D^l(R_e) = J Z(beta) J^T Z(alpha): one thread per (edge, row a).  (hamgnn_amd/so3.py:edge_wigner is the host twin.)
template <int L>
__device__ __forceinline__ void wigner_rows(const float4* __restrict__ ang, int64_t E, const float* __restrict__ J, const float* __restrict__ sgn,
                                            float* __restrict__ wig, int nW, int off) {
    constexpr int N = 2 * L + 1;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * N) return;
    const int64_t e = idx / N;
    const int a = (int)(idx - e * N);
    const float4 q = ang[e];
    float ca[L + 1], sa[L + 1], cb[L + 1], sb[L + 1];
    ca[0] = cb[0] = 1.f; sa[0] = sb[0] = 0.f;
#pragma unroll
    for (int m = 1; m <= L; ++m) {
        ca[m] = ca[m - 1] * q.x - sa[m - 1] * q.y; sa[m] = sa[m - 1] * q.x + ca[m - 1] * q.y;
        cb[m] = cb[m - 1] * q.z - sb[m - 1] * q.w; sb[m] = sb[m - 1] * q.z + cb[m - 1] * q.w;
    }
    float t1[N], t2[N];
    t1[L] = J[a * N + L];
#pragma unroll
    for (int m = 1; m <= L; ++m) {
        const float jp = J[a * N + L + m], jm = J[a * N + L - m], s = sgn[m] * sb[m];
        t1[L + m] = jp * cb[m] - s * jm;
        t1[L - m] = jm * cb[m] + s * jp;
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < N; ++b) acc = fmaf(t1[b], J[c * N + b], acc);
        t2[c] = acc;
    }
    float* __restrict__ o = wig + e * nW + off + a * N;
    o[L] = t2[L];
#pragma unroll
    for (int m = 1; m <= L; ++m) {
        const float s = sgn[m] * sa[m];
        o[L + m] = t2[L + m] * ca[m] - s * t2[L - m];
        o[L - m] = t2[L - m] * ca[m] + s * t2[L + m];
    }
}
template <int L>
__global__ void wigner_kernel(const float4* __restrict__ ang, int64_t E, const float* __restrict__ J, const float* __restrict__ sgn,
                              float* __restrict__ wig, int nW, int off) {
    wigner_rows<L>(ang, E, J, sgn, wig, nW, off);
}
// all l in ONE launch (blockIdx.y = l): small crystals, where seven extra launches are 4 % of a forward; blocks past E (2 l + 1) exit
__global__ void wigner_all_kernel(const float4* __restrict__ ang, int64_t E, const float* __restrict__ jtab, int nJ, float* __restrict__ wig, int nW) {
    const int l = blockIdx.y;
    int off = 0, soff = nJ;
    for (int k = 0; k < l; ++k) {
        off += (2 * k + 1) * (2 * k + 1);
        soff += k + 1;
    }
    const float* __restrict__ J = jtab + off;
    const float* __restrict__ sg = jtab + soff;
    switch (l) {
        case 0: wigner_rows<0>(ang, E, J, sg, wig, nW, off); break;
        case 1: wigner_rows<1>(ang, E, J, sg, wig, nW, off); break;
        case 2: wigner_rows<2>(ang, E, J, sg, wig, nW, off); break;
        case 3: wigner_rows<3>(ang, E, J, sg, wig, nW, off); break;
        case 4: wigner_rows<4>(ang, E, J, sg, wig, nW, off); break;
        case 5: wigner_rows<5>(ang, E, J, sg, wig, nW, off); break;
        case 6: wigner_rows<6>(ang, E, J, sg, wig, nW, off); break;
        case 7: wigner_rows<7>(ang, E, J, sg, wig, nW, off); break;
        default: break;
    }
}

extern "C" int hg_edge_geometry(const float* pos, const int64_t* edge_index, const float* nbr_shift, int64_t E, float cutoff,
                                int num_radial, int lmax_wig, const float* jtab, float* rbf, float* wig, float* edge_len,
                                float* ang_scratch, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (E <= 0) return 0;
    if (lmax_wig > 7) return hg_fail(-2, "hg_edge_geometry: lmax_wig > 7 not instantiated");
    hipStream_t st = (hipStream_t)stream;
    int nW = 0, nJ = 0;
    for (int l = 0; l <= lmax_wig; ++l) nW += (2 * l + 1) * (2 * l + 1);
    nJ = nW;
    float4* ang = reinterpret_cast<float4*>(ang_scratch);
    geometry_kernel<<<dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st>>>(pos, edge_index, nbr_shift, E, cutoff, num_radial, ang, rbf, edge_len);
    if (wig && E * (2 * lmax_wig + 1) <= 256 * 2048) {        // small crystal: every l in one launch
        wigner_all_kernel<<<dim3((unsigned)((E * (2 * lmax_wig + 1) + 255) / 256), (unsigned)(lmax_wig + 1)), 256, 0, st>>>(ang, E, jtab, nJ, wig, nW);
    } else if (wig) {
        int off = 0, soff = nJ;
        for (int l = 0; l <= lmax_wig; ++l) {
            const int N = 2 * l + 1;
            const unsigned grid = (unsigned)((E * N + 255) / 256);
            const float* J = jtab + off;
            const float* sg = jtab + soff;
            switch (l) {
                case 0: wigner_kernel<0><<<grid, 256, 0, st>>>(ang, E, J, sg, wig, nW, off); break;
                case 1: wigner_kernel<1><<<grid, 256, 0, st>>>(ang, E, J, sg, wig, nW, off); break;
                case 2: wigner_kernel<2><<<grid, 256, 0, st>>>(ang, E, J, sg, wig, nW, off); break;
                case 3: wigner_kernel<3><<<grid, 256, 0, st>>>(ang, E, J, sg, wig, nW, off); break;
                case 4: wigner_kernel<4><<<grid, 256, 0, st>>>(ang, E, J, sg, wig, nW, off); break;
                case 5: wigner_kernel<5><<<grid, 256, 0, st>>>(ang, E, J, sg, wig, nW, off); break;
                case 6: wigner_kernel<6><<<grid, 256, 0, st>>>(ang, E, J, sg, wig, nW, off); break;
                case 7: wigner_kernel<7><<<grid, 256, 0, st>>>(ang, E, J, sg, wig, nW, off); break;
            }
            off += N * N;
            soff += l + 1;
        }
    }
    return hg_check_launch("hg_edge_geometry");
}

// ------------------------------------------------------------------------------------------------ other radial bases
// rbf_func = "gaussian" (hamgnn/models/hamgnn_conv.py:123-125 -> GaussianSmearing, utils/basis_functions.py:211-224, start 0, stop =
// cutoff, no inner cutoff) x the cosine cutoff of RadialBasisEdgeEncoding (nn/embeddings.py:93-97), from the edge lengths that
// hg_edge_geometry wrote.  fp64 internally, rounded once.  `offsets` = the reference's fp32 centres (torch.linspace is not n * delta
// to the last bit), `delta` = offsets[1] - offsets[0] in fp32 (the reference's width): both from the host, so both sides use the same numbers.
__global__ void gaussian_basis_kernel(const float* __restrict__ len, int64_t E, float cutoff, const float* __restrict__ offsets, float delta, int R,
                                      float* __restrict__ rbf) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * R) return;
    const int64_t e = idx / R;
    const int n = (int)(idx - e * R);
    const double r = (double)len[e];
    const double off = (double)offsets[n];
    const double fc = (r < (double)cutoff) ? 0.5 * (cos(M_PI * r / (double)cutoff) + 1.0) : 0.0;
    const double d = r - off;
    rbf[idx] = (float)(exp(-0.5 * d * d / ((double)delta * (double)delta)) * fc);
}

extern "C" int hg_radial_basis(const float* edge_len, int64_t E, int kind, float cutoff, const float* offsets, float delta, int num_radial, float* rbf,
                               void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (E <= 0) return 0;
    if (kind != 1) return hg_fail(-2, "hg_radial_basis: kind 1 (gaussian) is built; bessel comes from hg_edge_geometry");
    if (num_radial < 2) return hg_fail(-2, "hg_radial_basis: num_radial >= 2");
    gaussian_basis_kernel<<<dim3((unsigned)((E * num_radial + 255) / 256)), 256, 0, (hipStream_t)stream>>>(edge_len, E, cutoff, offsets, delta, num_radial, rbf);
    return hg_check_launch("hg_radial_basis");
}

// ------------------------------------------------------------------------------------------------ radial hidden layers
// 64 edges per workgroup; activations and the current layer's weights in LDS; each thread owns a 4 (edges) x 4 (outputs)
// register tile per pass.  act(x) = cst * silu(x) (e3nn normalize2mom(silu)); weights already carry 1/sqrt(h_in).
#define RH_TE 64
__global__ __launch_bounds__(256) void radial_hidden_kernel(const float* __restrict__ rbf, int64_t E, const float* __restrict__ W,
                                                            int d0, int d1, int d2, int d3, int nlayers, float cst,
                                                            float* __restrict__ out, int maxd) {
    extern __shared__ float sm[];
    const int ld = maxd + 1;                                   // +1: conflict-free column access
    float* buf0 = sm;
    float* buf1 = sm + RH_TE * ld;
    float* wsm = sm + 2 * RH_TE * ld;                          // [di][dn]
    const int64_t e0 = (int64_t)blockIdx.x * RH_TE;
    const int dims[4] = {d0, d1, d2, d3};
    for (int i = threadIdx.x; i < RH_TE * d0; i += blockDim.x) {
        const int e = i / d0, k = i - e * d0;
        buf0[e * ld + k] = (e0 + e < E) ? rbf[(e0 + e) * d0 + k] : 0.f;
    }
    const float* w = W;
    float* in = buf0;
    float* ot = buf1;
    for (int l = 0; l < nlayers; ++l) {
        const int di = dims[l], dn = dims[l + 1];
        __syncthreads();
        for (int i = threadIdx.x; i < di * dn; i += blockDim.x) wsm[i] = w[i];
        __syncthreads();
        const int ng = (dn + 3) >> 2;                          // output groups of 4
        for (int tile = threadIdx.x; tile < (RH_TE / 4) * ng; tile += blockDim.x) {
            const int eg = tile / ng, og = tile - eg * ng;
            float acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
            for (int k = 0; k < di; ++k) {
                float xv[4], wv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) xv[a] = in[(4 * eg + a) * ld + k];
#pragma unroll
                for (int b = 0; b < 4; ++b) wv[b] = (4 * og + b < dn) ? wsm[k * dn + 4 * og + b] : 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(xv[a], wv[b], acc[a][b]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (4 * og + b < dn) {
                        const float v = acc[a][b];
                        ot[(4 * eg + a) * ld + 4 * og + b] = cst * v / (1.f + __expf(-v));
                    }
        }
        w += di * dn;
        float* t = in; in = ot; ot = t;
    }
    __syncthreads();
    const int dl = dims[nlayers];
    for (int i = threadIdx.x; i < RH_TE * dl; i += blockDim.x) {
        const int e = i / dl, j = i - e * dl;
        if (e0 + e < E) out[(e0 + e) * dl + j] = in[e * ld + j];
    }
}

// MFMA version for the shipped shape (64 radial functions -> 64 -> 64): edges are the MFMA columns (16 per wave tile), both weight
// matrices live in registers as A fragments (2 x 16 float4 per lane) for the whole persistent loop, and layer 1's C fragments
// feed layer 2 as B operands directly (permuted-K packing, as in tp_fused.hip): 128 v_mfma_f32_16x16x4_f32 per 16 edges, one
// coalesced read of the rbf row, one write of the hidden row.  (The LDS/VALU kernel above ran at 21 TFLOP/s, 0.64 ms per launch.)
typedef float rh_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rh_f4 rh_silu(rh_f4 v, float cst) {
    rh_f4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = cst * v[r] / (1.f + __expf(-v[r]));
    return o;
}
__global__ __launch_bounds__(256) void radial_hidden_mfma_kernel(const float* __restrict__ rbf, int64_t E, const float* __restrict__ W_all, float cst,
                                                                 float* __restrict__ out_all) {
    // blockIdx.y: which MLP of a batch of 64 -> 64 -> 64 weight generators that share the radial basis rows (hg_radial_hidden_multi: all
    // 13 generators of a 3-layer backbone in ONE launch -- the basis rows come from HBM once, the other readers hit L2 / Infinity Cache)
    const float* __restrict__ W = W_all + (int64_t)blockIdx.y * 8192;
    float* __restrict__ out = out_all + (int64_t)blockIdx.y * E * 64;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    // A fragments: lane (i, g), tile (rt, T), register q  <-  W[in = 16 T + 4 g + q][out = 16 rt + i]   (W row-major [in][out])
    rh_f4 a1[4][4], a2[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a1[rt][T][q] = W[(16 * T + 4 * g + q) * 64 + 16 * rt + i];
                a2[rt][T][q] = W[4096 + (16 * T + 4 * g + q) * 64 + 16 * rt + i];
            }
    const int64_t ntile = (E + 15) >> 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntile; t += (int64_t)gridDim.x * 4) {
        const int64_t e = t * 16 + i;
        const int64_t er = e < E ? e : E - 1;
        rh_f4 x[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) x[T] = *reinterpret_cast<const rh_f4*>(rbf + er * 64 + 16 * T + 4 * g);
        rh_f4 h1[4], h2[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            rh_f4 acc = (rh_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[rt][T][q], x[T][q], acc, 0, 0, 0);
            h1[rt] = rh_silu(acc, cst);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            rh_f4 acc = (rh_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[rt][T][q], h1[T][q], acc, 0, 0, 0);
            h2[rt] = rh_silu(acc, cst);
        }
        if (e < E) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) *reinterpret_cast<rh_f4*>(out + e * 64 + 16 * rt + 4 * g) = h2[rt];
        }
    }
}

extern "C" int hg_radial_hidden(const float* rbf, int64_t E, const float* weights, const int32_t* dims, int nlayers, float act_cst,
                                float* h_out, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (E <= 0) return 0;
    if (nlayers < 1 || nlayers > 3) return hg_fail(-2, "hg_radial_hidden: 1..3 hidden layers supported");
    if (nlayers == 2 && dims[0] == 64 && dims[1] == 64 && dims[2] == 64) {
        const int64_t nwg = (E + 63) / 64;
        const unsigned grid = (unsigned)(nwg < 2048 ? nwg : 2048);            // persistent: the weight fragments are loaded once per wave
        radial_hidden_mfma_kernel<<<dim3(grid), 256, 0, (hipStream_t)stream>>>(rbf, E, weights, act_cst, h_out);
        return hg_check_launch("hg_radial_hidden");
    }
    int d[4] = {0, 0, 0, 0}, maxd = 0, maxw = 0;
    for (int i = 0; i <= nlayers; ++i) { d[i] = dims[i]; if (d[i] > maxd) maxd = d[i]; }
    for (int i = 0; i < nlayers; ++i) if (d[i] * d[i + 1] > maxw) maxw = d[i] * d[i + 1];
    const size_t lds = (2 * RH_TE * (size_t)(maxd + 1) + (size_t)maxw) * sizeof(float);
    if (lds > 64 * 1024) return hg_fail(-2, "hg_radial_hidden: layer too wide for the LDS-resident kernel");
    radial_hidden_kernel<<<dim3((unsigned)((E + RH_TE - 1) / RH_TE)), 256, lds, (hipStream_t)stream>>>(rbf, E, weights, d[0], d[1], d[2], d[3], nlayers, act_cst, h_out, maxd);
    return hg_check_launch("hg_radial_hidden");
}

extern "C" int hg_radial_hidden_multi(const float* rbf, int64_t E, const float* weights, int nmlp, float act_cst, float* h_out, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (E <= 0 || nmlp <= 0) return 0;
    if (nmlp > 1024) return hg_fail(-2, "hg_radial_hidden_multi: at most 1024 generators per launch");
    const int64_t nwg = (E + 63) / 64;
    const unsigned grid = (unsigned)(nwg < 2048 ? nwg : 2048);
    radial_hidden_mfma_kernel<<<dim3(grid, (unsigned)nmlp), 256, 0, (hipStream_t)stream>>>(rbf, E, weights, act_cst, h_out);
    return hg_check_launch("hg_radial_hidden_multi");
}

// ------------------------------------------------------------------------------------------------ gather + frame rotation
// out_s[e][i][a][u] = sum_b D_e^{l_i}[a][b] x_s[idx_s[e]][i][b][u].  Work item = (edge, source, group of 4 channels of one irrep):
// 2l+1 float4 loads (component stride mulp), (2l+1)^2 x 4 FMAs against the edge's Wigner block staged in LDS (broadcast
// reads, amortised over the 4 channels), 2l+1 float4 stores.  The group table is sorted by l and a wave covers 4 consecutive
// groups x 16 (edge, source) pairs, so a wavefront runs one -- at block seams two -- of the <L> paths.  (The r1 kernel used one
// channel per lane in layout order: a wavefront then spanned every l of the row and executed all seven paths serially -- it
// was VALU-issue bound at 1.3 TB/s instead of HBM bound.)
// grp_tab: int32[ngroups][4] = {l, planar offset of (component 0, first channel), mulp, valid channels (1..4)}.
typedef float rg_f4 __attribute__((ext_vector_type(4)));
template <int L>
__device__ __forceinline__ void rotate_group(const float* __restrict__ Dl, const float* __restrict__ xin, float* __restrict__ xout,
                                             int base, int mulp, int transpose, int nvalid) {
    constexpr int N = 2 * L + 1;
    rg_f4 v[N];
#pragma unroll
    for (int b = 0; b < N; ++b) v[b] = *reinterpret_cast<const rg_f4*>(xin + base + b * mulp);
    const rg_f4 keep = (rg_f4){1.f, nvalid > 1 ? 1.f : 0.f, nvalid > 2 ? 1.f : 0.f, nvalid > 3 ? 1.f : 0.f};
#pragma unroll
    for (int a = 0; a < N; ++a) {
        rg_f4 acc = (rg_f4){0.f, 0.f, 0.f, 0.f};
        if (!transpose) {
#pragma unroll
            for (int b = 0; b < N; ++b) acc += Dl[a * N + b] * v[b];
        } else {
#pragma unroll
            for (int b = 0; b < N; ++b) acc += Dl[b * N + a] * v[b];
        }
        *reinterpret_cast<rg_f4*>(xout + base + a * mulp) = acc * keep;        // channel padding stays zero (zero-weight MFMA K-steps)
    }
}

#define RG_EB 16         // edges per workgroup: one frame staging + barrier per 16 x nsrc x ngroups work items
__global__ __launch_bounds__(256) void rotate_gather_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int64_t xs,
                                                            const int64_t* __restrict__ idx0, const int64_t* __restrict__ idx1,
                                                            const float* __restrict__ wig, int nW, const HgWigOff wo,
                                                            const int4* __restrict__ tab, int ngroups, int64_t E, int transpose,
                                                            float* __restrict__ out0, float* __restrict__ out1, int64_t os) {
    extern __shared__ float Dsm[];                             // [RG_EB][nW]
    const int64_t e0 = (int64_t)blockIdx.x * RG_EB;
    const int ne = (int)((E - e0) < RG_EB ? (E - e0) : RG_EB);
    for (int i = threadIdx.x; i < ne * nW; i += blockDim.x) Dsm[i] = wig[e0 * nW + i];
    __syncthreads();
    const int nsrc = x1 ? 2 : 1;
    const int npairs = RG_EB * nsrc;
    const int nsg = (ngroups + 3) >> 2;                        // super-groups of 4 consecutive groups
    for (int j = threadIdx.x; j < nsg * npairs * 4; j += blockDim.x) {
        const int sg = j / (npairs * 4);
        const int rem = j - sg * npairs * 4;
        const int pair = rem >> 2;
        const int gid = sg * 4 + (rem & 3);
        const int le = pair / nsrc;
        const int sidx = pair - le * nsrc;
        if (gid >= ngroups || le >= ne) continue;
        const int4 t = tab[gid];
        const int64_t e = e0 + le;
        const int64_t row = sidx ? (idx1 ? idx1[e] : e) : (idx0 ? idx0[e] : e);
        const float* __restrict__ xin = (sidx ? x1 : x0) + row * xs;
        float* __restrict__ xo = (sidx ? out1 : out0) + e * os;
        const float* __restrict__ Dl = Dsm + le * nW + wo.o[t.x];
        switch (t.x) {
            case 0: {
                const rg_f4 keep = (rg_f4){1.f, t.w > 1 ? 1.f : 0.f, t.w > 2 ? 1.f : 0.f, t.w > 3 ? 1.f : 0.f};
                *reinterpret_cast<rg_f4*>(xo + t.y) = *reinterpret_cast<const rg_f4*>(xin + t.y) * keep;
                break;
            }
            case 1: rotate_group<1>(Dl, xin, xo, t.y, t.z, transpose, t.w); break;
            case 2: rotate_group<2>(Dl, xin, xo, t.y, t.z, transpose, t.w); break;
            case 3: rotate_group<3>(Dl, xin, xo, t.y, t.z, transpose, t.w); break;
            case 4: rotate_group<4>(Dl, xin, xo, t.y, t.z, transpose, t.w); break;
            case 5: rotate_group<5>(Dl, xin, xo, t.y, t.z, transpose, t.w); break;
            case 6: rotate_group<6>(Dl, xin, xo, t.y, t.z, transpose, t.w); break;
            case 7: rotate_group<7>(Dl, xin, xo, t.y, t.z, transpose, t.w); break;      // su2 couplings of f-shell bases (gradient rows of the read-out)
            default: break;
        }
    }
}

extern "C" int hg_rotate_gather(const float* x0, const float* x1, int64_t x_stride, const int64_t* idx0, const int64_t* idx1,
                                const float* wig, int nW, const int32_t* wig_off, const int32_t* grp_tab, int ngroups, int64_t E,
                                int transpose, float* out0, float* out1, int64_t out_stride, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (E <= 0) return 0;
    if ((x_stride & 3) || (out_stride & 3)) return hg_fail(-2, "hg_rotate_gather: row strides must be multiples of 4 floats (planar rows)");
    HgWigOff wo;
    for (int i = 0; i < 8; ++i) wo.o[i] = wig_off[i];
    const size_t lds = sizeof(float) * (size_t)nW * RG_EB;
    if (lds > 64 * 1024) return hg_fail(-2, "hg_rotate_gather: Wigner row too wide for the LDS staging");
    rotate_gather_kernel<<<dim3((unsigned)((E + RG_EB - 1) / RG_EB)), 256, lds, (hipStream_t)stream>>>(
        x0, x1, x_stride, idx0, idx1, wig, nW, wo, (const int4*)grp_tab, ngroups, E, transpose, out0, out1, out_stride);
    return hg_check_launch("hg_rotate_gather");
}

// ------------------------------------------------------------------------------------------------ segmented node scatter
// Wavefront segmented reduce over the receiver CSR: one block per node; a lane owns one 16-byte column of the row, so a wavefront reads
// 1 KiB of each incoming message row with one coalesced float4 load; the edge list is walked four edges at a time (four independent
// index loads, then four row loads in flight) and summed in list order -- fixed order, hence bit-reproducible, unlike the atomics of
// torch_scatter.scatter (hamgnn/nn/convolution.py:147-149).
typedef float ss_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ msg, int64_t ms, const int64_t* __restrict__ rowptr,
                                                          const int64_t* __restrict__ perm, int Dp, float* __restrict__ out, int64_t os) {
    const int64_t n = blockIdx.x;
    const int64_t q0 = rowptr[n], q1 = rowptr[n + 1];
    for (int p = 4 * threadIdx.x; p < Dp; p += 4 * blockDim.x) {
        ss_f4 acc = (ss_f4){0.f, 0.f, 0.f, 0.f};
        int64_t q = q0;
        for (; q + 4 <= q1; q += 4) {
            const int64_t e0 = perm[q], e1 = perm[q + 1], e2 = perm[q + 2], e3 = perm[q + 3];
            const ss_f4 v0 = *reinterpret_cast<const ss_f4*>(msg + e0 * ms + p), v1 = *reinterpret_cast<const ss_f4*>(msg + e1 * ms + p);
            const ss_f4 v2 = *reinterpret_cast<const ss_f4*>(msg + e2 * ms + p), v3 = *reinterpret_cast<const ss_f4*>(msg + e3 * ms + p);
            acc = (((acc + v0) + v1) + v2) + v3;
        }
        for (; q < q1; ++q) acc += *reinterpret_cast<const ss_f4*>(msg + perm[q] * ms + p);
        *reinterpret_cast<ss_f4*>(out + n * os + p) = acc;
    }
}

extern "C" int hg_segment_sum(const float* msg, int64_t msg_stride, const int64_t* rowptr, const int64_t* perm, int64_t N, int Dp,
                              float* out, int64_t out_stride, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (N <= 0) return 0;
    if ((Dp & 3) || (msg_stride & 3) || (out_stride & 3) || ((reinterpret_cast<uintptr_t>(msg) | reinterpret_cast<uintptr_t>(out)) & 15))
        return hg_fail(-2, "hg_segment_sum: rows must be multiples of 4 floats and 16-byte aligned (planar rows are)");
    segment_sum_kernel<<<dim3((unsigned)N), 256, 0, (hipStream_t)stream>>>(msg, msg_stride, rowptr, perm, Dp, out, out_stride);
    return hg_check_launch("hg_segment_sum");
}

// ------------------------------------------------------------------------------------------------ gate / adds / layout
__device__ __forceinline__ float hg_act(float x, int id, const float* __restrict__ cst) {
    switch (id) {
        case 1: return cst[1] * ((x > 20.f ? x : log1pf(__expf(x))) - 0.6931471805599453f);     // shifted softplus
        case 2: return cst[2] * tanhf(x);
        case 3: return cst[3] * x / (1.f + __expf(-x));
        case 4: return cst[4] * fabsf(x);
        default: return x;
    }
}

// One wave per row, no block barriers.  Phase A: the row's DISTINCT activated scalars (the 0e / 0o scalars and the gate channels: 263 of
// the 1012 inputs for set-A) go through their activation once, into a wave-private LDS strip; phase B: every output element is a plain
// input or an activated scalar, times its gate's activated value -- look-ups only.  (The r1 kernel evaluated softplus = log1pf(expf)
// once per OUTPUT element, i.e. (2l+1) times per gate channel: the E-row gates of the head ran compute-bound at 1.8 TB/s.)
// act_tab: int32[nact][2] = {input index, act id};  out_tab: int32[Dout][2] = {source code, gate code}: source code = input index, or
// 0x40000000 | act slot; gate code = act slot or -1; source code -1 = structural zero.
#define HG_GATE_WAVES 4
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ x, int64_t xs, const int2* __restrict__ act_tab, int nact,
                                                   const int2* __restrict__ out_tab, int Dout, const float* __restrict__ cst, int64_t rows,
                                                   float* __restrict__ out, int64_t os) {
    extern __shared__ __attribute__((aligned(16))) int sm_i[];
    int2* __restrict__ s_out = reinterpret_cast<int2*>(sm_i);                           // [Dout]
    int2* __restrict__ s_act = s_out + Dout;                                            // [nact]
    float* __restrict__ s_val = reinterpret_cast<float*>(s_act + nact);                 // [HG_GATE_WAVES][nact]
    for (int i = threadIdx.x; i < Dout; i += blockDim.x) s_out[i] = out_tab[i];
    for (int i = threadIdx.x; i < nact; i += blockDim.x) s_act[i] = act_tab[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* __restrict__ av = s_val + wave * nact;
    float c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = cst[i];
    for (int64_t r = (int64_t)blockIdx.x * HG_GATE_WAVES + wave; r < rows; r += (int64_t)gridDim.x * HG_GATE_WAVES) {
        const float* __restrict__ xr = x + r * xs;
        for (int i = lane; i < nact; i += 64) {
            const int2 t = s_act[i];
            av[i] = hg_act(xr[t.x], t.y, c);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* __restrict__ orow = out + r * os;
#pragma unroll 4
        for (int p = lane; p < Dout; p += 64) {
            const int2 t = s_out[p];
            float v = 0.f;
            if (t.x >= 0) {
                v = (t.x & 0x40000000) ? av[t.x & 0x3fffffff] : xr[t.x];
                if (t.y >= 0) v *= av[t.y];
            }
            orow[p] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // the strip is rewritten by the next row
    }
}

extern "C" int hg_gate(const float* x, int64_t x_stride, const int32_t* act_tab, int nact, const int32_t* out_tab, int Dout, const float* consts,
                       int64_t rows, float* out, int64_t out_stride, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nact < 0 || Dout <= 0) return hg_fail(-2, "hg_gate: bad table sizes");
    const size_t lds = sizeof(int) * (2 * (size_t)Dout + 2 * (size_t)nact + (size_t)HG_GATE_WAVES * (size_t)nact);
    if (lds > 64 * 1024) return hg_fail(-2, "hg_gate: row too wide for the LDS tables");
    const int64_t want = (rows + HG_GATE_WAVES - 1) / HG_GATE_WAVES;
    const int64_t blocks = want < 256 * 8 ? want : 256 * 8;    // persistent: the tables are staged once per block
    gate_kernel<<<dim3((unsigned)blocks), 256, lds, (hipStream_t)stream>>>(x, x_stride, (const int2*)act_tab, nact, (const int2*)out_tab, Dout,
                                                                           consts, rows, out, out_stride);
    return hg_check_launch("hg_gate");
}

// Backward of the gate (data gradient): with out[p] = f(x[src]) (activated scalar), x[src] * act(x[gate]) (gated component),
//   gx[src]  = gy[p] * f'(x[src])                       activated scalar
//   gx[src]  = gy[p] * act(x[gate])                     gated component
//   gx[gate] = act'(x[gate]) * sum_p gy[p] * x[src(p)]  over the components the gate channel multiplies
// Same tables and the same wave-per-row structure as gate_kernel: activations and their derivatives once per row into wave-private LDS
// strips, the per-gate sums accumulate there (ds_add_f32 inside one wave: no barrier), input columns no output reads get 0.
__device__ __forceinline__ float hg_act_grad(float x, int id, const float* __restrict__ cst) {
    switch (id) {
        case 1: return cst[1] / (1.f + __expf(-x));                                             // d/dx softplus = sigmoid
        case 2: { const float t = tanhf(x); return cst[2] * (1.f - t * t); }
        case 3: { const float s = 1.f / (1.f + __expf(-x)); return cst[3] * s * (1.f + x * (1.f - s)); }
        case 4: return cst[4] * (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f));
        default: return 1.f;
    }
}
__global__ __launch_bounds__(256) void gate_backward_kernel(const float* __restrict__ x, int64_t xs, const float* __restrict__ gy, int64_t gs,
                                                            const int2* __restrict__ act_tab, int nact, const int2* __restrict__ out_tab, int Dout,
                                                            const float* __restrict__ cst, int64_t rows, int Din, float* __restrict__ gx, int64_t gxs) {
    extern __shared__ __attribute__((aligned(16))) int sm_i[];
    int2* __restrict__ s_out = reinterpret_cast<int2*>(sm_i);                           // [Dout]
    int2* __restrict__ s_act = s_out + Dout;                                            // [nact]
    float* __restrict__ s_val = reinterpret_cast<float*>(s_act + nact);                 // [HG_GATE_WAVES][3 nact + Din]: act, act', gate sums, the gx row
    for (int i = threadIdx.x; i < Dout; i += blockDim.x) s_out[i] = out_tab[i];
    for (int i = threadIdx.x; i < nact; i += blockDim.x) s_act[i] = act_tab[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* __restrict__ av = s_val + wave * (3 * nact + Din);
    float* __restrict__ dv = av + nact;
    float* __restrict__ sv = dv + nact;
    float* __restrict__ row = sv + nact;                       // the gradient row is assembled in LDS (every column written once, then
    float c[5];                                                // gate columns accumulate), and leaves as one coalesced copy
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = cst[i];
    for (int64_t r = (int64_t)blockIdx.x * HG_GATE_WAVES + wave; r < rows; r += (int64_t)gridDim.x * HG_GATE_WAVES) {
        const float* __restrict__ xr = x + r * xs;
        const float* __restrict__ gr = gy + r * gs;
        for (int i = lane; i < Din; i += 64) row[i] = 0.f;
        for (int i = lane; i < nact; i += 64) {
            const int2 t = s_act[i];
            av[i] = hg_act(xr[t.x], t.y, c);
            dv[i] = hg_act_grad(xr[t.x], t.y, c);
            sv[i] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int p = lane; p < Dout; p += 64) {
            const int2 t = s_out[p];
            if (t.x < 0) continue;
            const float g = gr[p];
            if (t.x & 0x40000000) {                            // activated scalar: slot -> its input column
                const int slot = t.x & 0x3fffffff;
                float v = g * dv[slot];
                if (t.y >= 0) {                                // (an activated scalar that is gated as well: not produced by the planner)
                    atomicAdd(&sv[t.y], g * av[slot]);
                    v *= av[t.y];
                }
                row[s_act[slot].x] = v;
            } else if (t.y >= 0) {                             // gated component
                row[t.x] = g * av[t.y];
                atomicAdd(&sv[t.y], g * xr[t.x]);
            } else {
                row[t.x] = g;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < nact; i += 64) row[s_act[i].x] += sv[i] * dv[i];          // gate channels: act'(x) * sum (plain scalars: + 0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* __restrict__ orow = gx + r * gxs;
        for (int i = lane; i < Din; i += 64) orow[i] = row[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                       // the strips are rewritten by the next row
    }
}

extern "C" int hg_gate_backward(const float* x, int64_t x_stride, const float* gy, int64_t gy_stride, const int32_t* act_tab, int nact,
                                const int32_t* out_tab, int Dout, const float* consts, int64_t rows, int Din, float* gx, int64_t gx_stride,
                                void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nact < 0 || Dout <= 0 || Din <= 0) return hg_fail(-2, "hg_gate_backward: bad table sizes");
    const size_t lds = sizeof(int) * (2 * (size_t)Dout + 2 * (size_t)nact + (size_t)HG_GATE_WAVES * (3 * (size_t)nact + (size_t)Din));
    if (lds > 64 * 1024) return hg_fail(-2, "hg_gate_backward: row too wide for the LDS tables");
    const int64_t want = (rows + HG_GATE_WAVES - 1) / HG_GATE_WAVES;
    const int64_t blocks = want < 256 * 8 ? want : 256 * 8;
    gate_backward_kernel<<<dim3((unsigned)blocks), 256, lds, (hipStream_t)stream>>>(x, x_stride, gy, gy_stride, (const int2*)act_tab, nact,
                                                                                    (const int2*)out_tab, Dout, consts, rows, Din, gx, gx_stride);
    return hg_check_launch("hg_gate_backward");
}

// e3nn NormActivation of a ResidualBlock with nonlinearity_type = "norm" (hamgnn/nn/interaction_blocks.py:311-330: scalar nonlinearity ssp AS GIVEN,
// normalize = True, epsilon = 1e-8, no bias) on planar rows: every irrep copy (channel u of block b: components x[off + a * mulp + u], a < 2 l + 1) is
// scaled by ssp(n) / n with n = sqrt(max(sum_a x_a^2, eps^2)).  chan_tab int32[nchan][2] = {offset of the channel's first component, component
// stride (mulp) | ncomp << 16}; the channel-padding slots of a planar row (no table entry) are written as zeros.  One thread per (row, channel).
__device__ __forceinline__ float hg_ssp(float x) { return (x > 20.f ? x : log1pf(__expf(x))) - 0.69314718055994531f; }
__global__ void norm_act_kernel(const float* __restrict__ x, int64_t xs, const int2* __restrict__ chan_tab, int nchan, float eps2, int64_t rows,
                                float* __restrict__ out, int64_t os) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nchan) return;
    const int64_t r = i / nchan;
    const int2 t = chan_tab[(int)(i - r * nchan)];
    const int st = t.y & 0xffff, nc = t.y >> 16;
    const float* __restrict__ xr = x + r * xs + t.x;
    float n2 = 0.f;
    for (int a = 0; a < nc; ++a) n2 = fmaf(xr[a * st], xr[a * st], n2);
    n2 = n2 < eps2 ? eps2 : n2;
    const float n = sqrtf(n2), sc = hg_ssp(n) / n;
    float* __restrict__ o = out + r * os + t.x;
    for (int a = 0; a < nc; ++a) o[a * st] = sc * xr[a * st];
}

extern "C" int hg_norm_act(const float* x, int64_t x_stride, const int32_t* chan_tab, int nchan, float eps, int64_t rows, float* out,
                           int64_t out_stride, int D, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nchan <= 0 || D <= 0) return hg_fail(-2, "hg_norm_act: bad table sizes");
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)rows * (size_t)out_stride, (hipStream_t)stream) != hipSuccess)      // channel-padding slots
        return hg_fail(-3, "hg_norm_act: memset failed");
    const int64_t n = rows * nchan;
    norm_act_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, x_stride, (const int2*)chan_tab, nchan, eps * eps, rows, out, out_stride);
    return hg_check_launch("hg_norm_act");
}

// its data gradient: y_a = s(n) / n * x_a  =>  gx_a = s / n * gy_a + x_a (sum_b gy_b x_b) (s'(n) - s / n) / n^2   (n > eps; below the clamp n is a
// constant and only the first term is left); s = ssp, s' = sigmoid
__global__ void norm_act_backward_kernel(const float* __restrict__ x, int64_t xs, const float* __restrict__ gy, int64_t gs, const int2* __restrict__ chan_tab,
                                         int nchan, float eps2, int64_t rows, float* __restrict__ gx, int64_t gxs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nchan) return;
    const int64_t r = i / nchan;
    const int2 t = chan_tab[(int)(i - r * nchan)];
    const int st = t.y & 0xffff, nc = t.y >> 16;
    const float* __restrict__ xr = x + r * xs + t.x;
    const float* __restrict__ gr = gy + r * gs + t.x;
    float n2 = 0.f, dot = 0.f;
    for (int a = 0; a < nc; ++a) {
        n2 = fmaf(xr[a * st], xr[a * st], n2);
        dot = fmaf(gr[a * st], xr[a * st], dot);
    }
    const bool clamped = n2 < eps2;
    n2 = clamped ? eps2 : n2;
    const float n = sqrtf(n2), s = hg_ssp(n), sc = s / n;
    const float k = clamped ? 0.f : dot * (1.f / (1.f + __expf(-n)) - sc) / n2;
    float* __restrict__ o = gx + r * gxs + t.x;
    for (int a = 0; a < nc; ++a) o[a * st] = fmaf(k, xr[a * st], sc * gr[a * st]);
}

extern "C" int hg_norm_act_backward(const float* x, int64_t x_stride, const float* gy, int64_t gy_stride, const int32_t* chan_tab, int nchan, float eps,
                                    int64_t rows, float* gx, int64_t gx_stride, int D, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nchan <= 0 || D <= 0) return hg_fail(-2, "hg_norm_act_backward: bad table sizes");
    if (hipMemsetAsync(gx, 0, sizeof(float) * (size_t)rows * (size_t)gx_stride, (hipStream_t)stream) != hipSuccess)
        return hg_fail(-3, "hg_norm_act_backward: memset failed");
    const int64_t n = rows * nchan;
    norm_act_backward_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, x_stride, gy, gy_stride, (const int2*)chan_tab, nchan,
                                                                                               eps * eps, rows, gx, gx_stride);
    return hg_check_launch("hg_norm_act_backward");
}

__global__ void add_rows_kernel(const float* __restrict__ a, int64_t sa, const float* __restrict__ b, int64_t sb,
                                const float* __restrict__ c, int64_t sc, int64_t rows, int D, float* __restrict__ out, int64_t so) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int64_t r = i / D;
    const int p = (int)(i - r * D);
    float v = a[r * sa + p] + b[r * sb + p];
    if (c) v += c[r * sc + p];
    out[r * so + p] = v;
}

extern "C" int hg_add_rows(const float* a, int64_t sa, const float* b, int64_t sb, const float* c, int64_t sc, int64_t rows, int D,
                           float* out, int64_t so, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    add_rows_kernel<<<dim3((unsigned)((rows * D + 255) / 256)), 256, 0, (hipStream_t)stream>>>(a, sa, b, sb, c, sc, rows, D, out, so);
    return hg_check_launch("hg_add_rows");
}

__global__ void to_planar_kernel(const float* __restrict__ x, int64_t rows, int D, const int* __restrict__ map, float* __restrict__ out, int Dp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int64_t r = i / D;
    const int k = (int)(i - r * D);
    out[r * Dp + map[k]] = x[i];
}
__global__ void from_planar_kernel(const float* __restrict__ xp, int64_t rows, int Dp, const int* __restrict__ map, float* __restrict__ out, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int64_t r = i / D;
    const int k = (int)(i - r * D);
    out[i] = map[k] >= 0 ? xp[r * Dp + map[k]] : 0.f;            // -1: structural zero (column gathers of the gradient layouts)
}
extern "C" int hg_to_planar(const float* x, int64_t rows, int D, const int32_t* map, float* out, int Dp, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)rows * Dp, (hipStream_t)stream);
    to_planar_kernel<<<dim3((unsigned)((rows * D + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, rows, D, map, out, Dp);
    return hg_check_launch("hg_to_planar");
}
extern "C" int hg_from_planar(const float* xp, int64_t rows, int Dp, const int32_t* map, float* out, int D, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    from_planar_kernel<<<dim3((unsigned)((rows * D + 255) / 256)), 256, 0, (hipStream_t)stream>>>(xp, rows, Dp, map, out, D);
    return hg_check_launch("hg_from_planar");
}

__global__ void embed_lookup_kernel(const float* __restrict__ Ta, const float* __restrict__ Tb, const int64_t* __restrict__ z,
                                    const int64_t* __restrict__ ia, const int64_t* __restrict__ ib, int64_t rows, int T, int Tp,
                                    float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * Tp) return;
    const int64_t r = i / Tp;
    const int t = (int)(i - r * Tp);
    float v = 0.f;
    if (t < T) {
        v = Ta[z[ia ? ia[r] : r] * T + t];
        if (Tb) v += Tb[z[ib[r]] * T + t];
    }
    out[i] = v;
}
extern "C" int hg_embed_lookup(const float* Ta, const float* Tb, const int64_t* z, const int64_t* idx_a, const int64_t* idx_b,
                               int64_t rows, int T, int Tp, float* out, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    embed_lookup_kernel<<<dim3((unsigned)((rows * Tp + 255) / 256)), 256, 0, (hipStream_t)stream>>>(Ta, Tb, z, idx_a, idx_b, rows, T, Tp, out);
    return hg_check_launch("hg_embed_lookup");
}


// ------------------------------------------------------------------------------------------------ measurement aid (bench.py roofline)
// What the fp32 matrix pipe sustains on THIS device at THIS moment: every wave issues `iters` x 8 v_mfma_f32_16x16x4_f32 on eight independent
// accumulators, operands from the caller's (random) buffer, two waves per SIMD on every CU -- nothing else in the loop.  The chip clocks to its
// power budget on non-trivial data (MI355X_MICROARCH.md, DVFS), so the nominal 157.3 TFLOP/s is quoted beside this number, not replaced by it.
__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
    rh_f4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = in[(tid * 16 + k) & 0xffff];
        b[k] = in[(tid * 16 + 8 + k) & 0xffff];
        acc[k] = (rh_f4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[(k + 3) & 7], acc[k], 0, 0, 0);
    }
    rh_f4 s = acc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += acc[k];
    out[tid] = s[0] + s[1] + s[2] + s[3];
}

// ---- split-half-precision twins of the radial weights (csrc/tp_is.hip: radial scale on v_mfma_f32_16x16x32_f16; plan/program.py:w3_split_fill) refilled ON THE DEVICE after a
// device-side repack (training): pair k = the two fp32 weights w[se[k]], w[so[k]] that share a dword of the twin; hi = f16(x 2^s), lo = f16((x 2^s - hi) 2^11), written as packed
// pairs at dh[k] / dl[k]; max |x 2^s| into maxabs (non-negative floats order like their bit patterns) for the host's range check.  One launch per program (the torch-op version: 18).
__global__ __launch_bounds__(256) void w3_split_refill_kernel(float* __restrict__ w, const int64_t* __restrict__ se, const int64_t* __restrict__ so, const int64_t* __restrict__ dh,
                                                              const int64_t* __restrict__ dl, int64_t n, float scale, float lo_scale, float* __restrict__ maxabs) {
    float m = 0.f;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const float xe = w[se[k]] * scale, xo = w[so[k]] * scale;
        const _Float16 he = (_Float16)xe, ho = (_Float16)xo;
        const _Float16 le = (_Float16)((xe - (float)he) * lo_scale), lo = (_Float16)((xo - (float)ho) * lo_scale);
        unsigned short a, b;
        __builtin_memcpy(&a, &he, 2); __builtin_memcpy(&b, &ho, 2);
        reinterpret_cast<unsigned*>(w)[dh[k]] = (unsigned)a | ((unsigned)b << 16);
        __builtin_memcpy(&a, &le, 2); __builtin_memcpy(&b, &lo, 2);
        reinterpret_cast<unsigned*>(w)[dl[k]] = (unsigned)a | ((unsigned)b << 16);
        m = fmaxf(m, fmaxf(fabsf(xe), fabsf(xo)));
    }
    if (!(m == m)) m = __int_as_float(0x7f800000);              // NaN weights: beyond any range
    atomicMax(reinterpret_cast<unsigned*>(maxabs), __float_as_uint(m));
}

extern "C" int hg_w3_split_refill(float* weights, const int64_t* src_even, const int64_t* src_odd, const int64_t* dst_hi, const int64_t* dst_lo, int64_t npairs,
                                  float scale, float lo_scale, float* maxabs, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (npairs <= 0) return 0;
    if (!weights || !src_even || !src_odd || !dst_hi || !dst_lo || !maxabs) return hg_fail(-2, "hg_w3_split_refill: null pointer");
    if (hipMemsetAsync(maxabs, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return hg_fail(-3, "hg_w3_split_refill: memset failed");
    const int64_t blocks = (npairs + 255) / 256;
    w3_split_refill_kernel<<<dim3((unsigned)(blocks < 1024 ? blocks : 1024)), 256, 0, (hipStream_t)stream>>>(weights, src_even, src_odd, dst_hi, dst_lo, npairs, scale, lo_scale, maxabs);
    return hg_check_launch("hg_w3_split_refill");
}

extern "C" int hg_mfma_probe(const float* in65536, float* out, int nblocks, int iters, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (nblocks <= 0 || iters <= 0) return hg_fail(-2, "hg_mfma_probe: nblocks and iters must be positive");
    mfma_probe_kernel<<<dim3((unsigned)nblocks), 256, 0, (hipStream_t)stream>>>(in65536, out, iters);
    return hg_check_launch("hg_mfma_probe");
}

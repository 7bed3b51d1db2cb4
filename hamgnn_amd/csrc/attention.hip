// attention.hip -- the two node-side kernels of AttentionBlockE3 (gfx950).  Hand-written HIP.
//
// Reference: hamgnn/nn/attention.py:91-164 (AttentionAggregation), :337-350 (AttentionBlockE3.forward), attention_utils.py:17-120.
//   logits[e, h] = cut(|r_e|) / sqrt(d_head) * < K[receiver_e] | K[sender_e] >_head h      (key AND query come from linear_key, :339-340)
//   w[e, h]      = exp(logits - max over the edges INTO the same node) / (sum + 1e-16)     (torch_geometric.utils.softmax)
//   out[n, col]  = sum over the edges e into n, in CSR order, of  w[e, head(col)] * V[e, col]
// Both are HBM-bound row streams over planar rows (component-major, channels padded to 4: a head is a channel range of every irrep
// block, `head_tab[col]` = its index or -1 for padding columns):
//   hg_attn_logits     one wave per edge: two gathered node rows (L2 / Infinity-Cache resident: N rows for E = ~80 N edges), per-head
//                      partial dot products per lane, one butterfly reduction per head; writes H floats per edge.
//   hg_attn_aggregate  one workgroup per receiver node: soft-max statistics of its <= few hundred incoming logits in LDS, then ONE pass
//                      over the value rows V[e, :] (the [E, Dp] output of the fused value MessagePackBlock -- the bytes that matter:
//                      Dp * 4 B per edge, read exactly once, float4 per thread, four rows in flight), fixed summation order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

#define AT_MAXH 8
typedef float at_f4 __attribute__((ext_vector_type(4)));

// exp(-1/x) for x > 0, else 0 (e3nn.math.soft_unit_step), x = cut_param * (1 - r / r_cut)  (hamgnn/utils/cutoff_functions.py:65-100)
__device__ __forceinline__ float at_soft_cut(float r, float cut_param, float cutoff) {
    const float x = cut_param * (1.f - r / cutoff);
    return x > 0.f ? expf(-1.f / x) : 0.f;
}

__global__ __launch_bounds__(256) void attn_logits_kernel(const float* __restrict__ K, int64_t ks, const int64_t* __restrict__ src,
                                                          const int64_t* __restrict__ dst, const float* __restrict__ length,
                                                          const int* __restrict__ head_tab, int Dp, int H, const float* __restrict__ cut_param,
                                                          float cutoff, float scale, int64_t E, float* __restrict__ logits) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float cp = cut_param[0];
    for (int64_t e = (int64_t)blockIdx.x * 4 + wave; e < E; e += (int64_t)gridDim.x * 4) {
        const float* __restrict__ a = K + src[e] * ks;
        const float* __restrict__ b = K + dst[e] * ks;
        float acc[AT_MAXH];
#pragma unroll
        for (int h = 0; h < AT_MAXH; ++h) acc[h] = 0.f;
        for (int p = 4 * lane; p < Dp; p += 256) {
            const at_f4 va = *reinterpret_cast<const at_f4*>(a + p), vb = *reinterpret_cast<const at_f4*>(b + p);
            const int4 hd = *reinterpret_cast<const int4*>(head_tab + p);
            const int hh[4] = {hd.x, hd.y, hd.z, hd.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pr = va[j] * vb[j];
#pragma unroll
                for (int h = 0; h < AT_MAXH; ++h) acc[h] += hh[j] == h ? pr : 0.f;
            }
        }
#pragma unroll
        for (int h = 0; h < AT_MAXH; ++h) {
            if (h < H) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc[h] += __shfl_xor(acc[h], o, 64);
            }
        }
        if (lane == 0) {
            const float w = at_soft_cut(length[e], cp, cutoff) * scale;
#pragma unroll
            for (int h = 0; h < AT_MAXH; ++h)
                if (h < H) logits[e * H + h] = w * acc[h];
        }
    }
}

#define AT_CH 128                 // edges per weight chunk in LDS
#define AT_WS (AT_MAXH + 1)       // weight row stride: slot AT_MAXH holds 0 (padding columns)

__global__ __launch_bounds__(256) void attn_aggregate_kernel(const float* __restrict__ logits, int H, const float* __restrict__ V, int64_t vs,
                                                             const int64_t* __restrict__ rowptr, const int64_t* __restrict__ perm,
                                                             const int* __restrict__ head_tab, int Dp, float* __restrict__ out, int64_t os) {
    __shared__ float s_red[256];
    __shared__ float s_m[AT_MAXH], s_inv[AT_MAXH];
    __shared__ float s_w[AT_CH * AT_WS];
    __shared__ int64_t s_e[AT_CH];
    const int64_t n = blockIdx.x;
    const int64_t q0 = rowptr[n], q1 = rowptr[n + 1];
    const int t = threadIdx.x, h = t & (AT_MAXH - 1), sub = t >> 3;                    // 32 threads per head slot
    // ---- soft-max statistics per head over the node's incoming edges (max, then sum of exp), torch_geometric.utils.softmax
    float mx = -INFINITY;
    if (h < H)
        for (int64_t q = q0 + sub; q < q1; q += 32) mx = fmaxf(mx, logits[perm[q] * H + h]);
    s_red[t] = mx;
    __syncthreads();
    if (t < AT_MAXH) {
        float m = -INFINITY;
        for (int k = 0; k < 32; ++k) m = fmaxf(m, s_red[k * AT_MAXH + t]);
        s_m[t] = m;
    }
    __syncthreads();
    float sm = 0.f;
    if (h < H) {
        const float m = s_m[h];
        for (int64_t q = q0 + sub; q < q1; q += 32) sm += expf(logits[perm[q] * H + h] - m);
    }
    s_red[t] = sm;
    __syncthreads();
    if (t < AT_MAXH) {
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += s_red[k * AT_MAXH + t];
        s_inv[t] = 1.f / (s + 1e-16f);
    }
    __syncthreads();
    // ---- weighted sum of the value rows, columns over the threads (float4 each), edges in CSR order
    for (int p0 = 0; p0 < Dp; p0 += 1024) {
        const int p = p0 + 4 * t;
        const bool active = p < Dp;
        int hc[4] = {AT_MAXH, AT_MAXH, AT_MAXH, AT_MAXH};
        if (active) {
            const int4 hd = *reinterpret_cast<const int4*>(head_tab + p);
            hc[0] = hd.x < 0 ? AT_MAXH : hd.x, hc[1] = hd.y < 0 ? AT_MAXH : hd.y, hc[2] = hd.z < 0 ? AT_MAXH : hd.z, hc[3] = hd.w < 0 ? AT_MAXH : hd.w;
        }
        at_f4 acc = (at_f4){0.f, 0.f, 0.f, 0.f};
        for (int64_t c0 = q0; c0 < q1; c0 += AT_CH) {
            const int nc = (int)((q1 - c0) < AT_CH ? (q1 - c0) : AT_CH);
            __syncthreads();                                   // previous chunk fully consumed
            for (int i = t; i < nc * AT_WS; i += 256) {
                const int qi = i / AT_WS, hh = i - qi * AT_WS;
                s_w[i] = hh < H ? expf(logits[perm[c0 + qi] * H + hh] - s_m[hh]) * s_inv[hh] : 0.f;
            }
            for (int i = t; i < nc; i += 256) s_e[i] = perm[c0 + i];
            __syncthreads();
            if (active) {
                const float* __restrict__ vp = V + p;
                int qi = 0;
                for (; qi + 4 <= nc; qi += 4) {
                    const at_f4 v0 = *reinterpret_cast<const at_f4*>(vp + s_e[qi] * vs), v1 = *reinterpret_cast<const at_f4*>(vp + s_e[qi + 1] * vs);
                    const at_f4 v2 = *reinterpret_cast<const at_f4*>(vp + s_e[qi + 2] * vs), v3 = *reinterpret_cast<const at_f4*>(vp + s_e[qi + 3] * vs);
                    const float* __restrict__ w = s_w + qi * AT_WS;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j] = fmaf(w[hc[j]], v0[j], acc[j]);
                        acc[j] = fmaf(w[AT_WS + hc[j]], v1[j], acc[j]);
                        acc[j] = fmaf(w[2 * AT_WS + hc[j]], v2[j], acc[j]);
                        acc[j] = fmaf(w[3 * AT_WS + hc[j]], v3[j], acc[j]);
                    }
                }
                for (; qi < nc; ++qi) {
                    const at_f4 v0 = *reinterpret_cast<const at_f4*>(vp + s_e[qi] * vs);
                    const float* __restrict__ w = s_w + qi * AT_WS;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[hc[j]], v0[j], acc[j]);
                }
            }
        }
        if (active) *reinterpret_cast<at_f4*>(out + n * os + p) = acc;
    }
}

extern "C" int hg_attn_logits(const float* K, int64_t k_stride, const int64_t* src, const int64_t* dst, const float* length,
                              const int32_t* head_tab, int Dp, int H, const float* cut_param, float cutoff, float scale, int64_t E,
                              float* logits, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (E <= 0) return 0;
    if (H < 1 || H > AT_MAXH) return hg_fail(-2, "hg_attn_logits: 1..8 heads");
    if ((Dp & 3) || (k_stride & 3) || (reinterpret_cast<uintptr_t>(K) & 15) || (reinterpret_cast<uintptr_t>(head_tab) & 15))
        return hg_fail(-2, "hg_attn_logits: rows must be multiples of 4 floats and 16-byte aligned (planar rows are)");
    const int64_t blocks = (E + 3) / 4;
    attn_logits_kernel<<<dim3((unsigned)(blocks < 65536 ? blocks : 65536)), 256, 0, (hipStream_t)stream>>>(
        K, k_stride, src, dst, length, head_tab, Dp, H, cut_param, cutoff, scale, E, logits);
    return hg_check_launch("hg_attn_logits");
}

extern "C" int hg_attn_aggregate(const float* logits, int H, const float* V, int64_t v_stride, const int64_t* rowptr, const int64_t* perm,
                                 const int32_t* head_tab, int64_t N, int Dp, float* out, int64_t out_stride, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (N <= 0) return 0;
    if (H < 1 || H > AT_MAXH) return hg_fail(-2, "hg_attn_aggregate: 1..8 heads");
    if ((Dp & 3) || (v_stride & 3) || (out_stride & 3) ||
        ((reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(head_tab)) & 15))
        return hg_fail(-2, "hg_attn_aggregate: rows must be multiples of 4 floats and 16-byte aligned (planar rows are)");
    attn_aggregate_kernel<<<dim3((unsigned)N), 256, 0, (hipStream_t)stream>>>(logits, H, V, v_stride, rowptr, perm, head_tab, Dp, out, out_stride);
    return hg_check_launch("hg_attn_aggregate");
}

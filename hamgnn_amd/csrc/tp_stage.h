// tp_stage.h -- operand staging (gather + frame rotation / LDS-DMA), epilogue (un-rotation + planar store) and the launch-argument block
// of the input-stationary edge kernel csrc/tp_is.hip (dynamic work claiming, register-chained row-tile chunks).
// Hand-written HIP for gfx950 (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// the 16 edges' hidden rows of a phase's radial MLP as B operands of v_mfma_f32_16x16x32_f16: x = hi + lo (csrc/tp_is.hip, plan/program.py:w3_split_fill)
struct IsHidden { f16x8 hi[2], lo[2]; };
#define IS_SPLIT_H_EXP 6         // the hidden rows are scaled by 2^6 before the split (= plan.SPLIT_H_EXP: small activations stay normal halves, |h| < 1023 finite)
#define IS_SPLIT_H_SCALE 64.f

#ifndef IS_NW
#define IS_NW 4                  // waves of a workgroup (all on the same 16 edges); plan.py:IS_WAVES.  6 = three waves per SIMD (experiment)
#endif
#define IS_NT (64 * IS_NW)
#ifndef IS_NW_LITE
#define IS_NW_LITE 8             // lite_mode programs (tp_is_kernel<SPLIT, LITE = true>): 8 waves per workgroup = four waves per SIMD at 128 VGPRs;
#endif                           // plan.py:IS_WAVES_LITE

struct IsArgs {
    const float* src[4];
    int64_t sstride[4];
    const float* h2[2];
    int hidden;
    const float* wig;
    int nW;
    int wig_off[8];
    float* out;
    int64_t ostride;
    int64_t rows;
    int nseg;                    // single-part launches: the whole program (split launches read these per part, see IS_PART_I32)
    int nphase;
    int trash_off;               // float offsets inside the workgroup's LDS
    int stage_off;
    int ctr_off;                 // work-claim counter (one dword)
    int rowtab_off;              // row table of GEMM2's output rows (ints): LDS float offset of every row's centre column, see plan.IsSchedule
    int rowtab_begin;            // first entry / entries of this part in the global table
    int rowtab_len;
    int tile_shift;              // split launches: this wave's private tile copy (floats added to every tile offset)
    int s_split;                 // the program carries the split-half-precision twins of its W3 blocks (part record [12]): radial scales on the f16 matrix pipe
    float s_scale;               // 2^-(sw + sh): undoes the scalings of the twins (part record [13]) and of the hidden rows
    const int64_t* idx[4];       // per source slot: row gather (NULL: row = edge)
    int rot_mask;                // bit i: source i holds GLOBAL-frame rows that are rotated into the edge frame while staged
    const int64_t* eperm;        // tile slot -> edge (NULL: identity).  Receiver-major order for launches whose output is scattered onto the receivers
    const int32_t* run_id;       // fused scatter (NULL: one output row per edge): slot -> index of its run of equal receivers inside the tile = the
                                 // output row that takes the run's sum (convolution.py:147-149 as a segmented reduce in the epilogue); -1: no output
};

// Fused node scatter: the 16 slots of a tile lie along a DPP row (lane = 16 g + slot).  seg_masks: for the Hillis-Steele steps d = 1, 2, 4, 8 of a
// segmented inclusive scan, 1.0 where slot el adds the partial sum of slot el - d (no run head in (el - d, el]), else 0.0; `last`: the slot closes
// its run (its lane holds the run's sum after the scan).  Fixed order of additions: bit-reproducible, unlike the reference's atomics.
struct IsScan {
    float m[4];
    bool last;
    int row;
};
template <int CTRL>
__device__ __forceinline__ int is_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float is_dpp_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ IsScan is_scan_setup(int rid, int el) {
    IsScan sc;
    sc.row = rid;
    const int prev = is_dpp_i<0x111>(rid), next = is_dpp_i<0x101>(rid);          // row_shr:1 / row_shl:1 (0 beyond the row's ends)
    int f = (el == 0 || prev != rid) ? 1 : 0;                                     // run head
    sc.last = el == 15 || next != rid;
    sc.m[0] = f ? 0.f : 1.f;
    int f1 = f | (el < 1 ? 1 : is_dpp_i<0x111>(f));
    sc.m[1] = f1 ? 0.f : 1.f;
    int f2 = f1 | (el < 2 ? 1 : is_dpp_i<0x112>(f1));
    sc.m[2] = f2 ? 0.f : 1.f;
    int f4 = f2 | (el < 4 ? 1 : is_dpp_i<0x114>(f2));
    sc.m[3] = f4 ? 0.f : 1.f;
    return sc;
}
__device__ __forceinline__ float is_seg_scan(float x, const IsScan& sc) {
    x = fmaf(sc.m[0], is_dpp_f<0x111>(x), x);
    x = fmaf(sc.m[1], is_dpp_f<0x112>(x), x);
    x = fmaf(sc.m[2], is_dpp_f<0x114>(x), x);
    x = fmaf(sc.m[3], is_dpp_f<0x118>(x), x);
    return x;
}

#define SEG_UNROTATE 1
#define SEG_ATOMIC 2             // (split launches only) the segment is one of TWO copies that share an output block: each adds its half of the items' sum
                                 // (plan.split_heavy_segments: the heaviest output irreps of a small crystal's launch on two workgroups; the rows are zero-filled by the host)

__device__ __forceinline__ const float* is_pick_src(const IsArgs& A, int i) {
    return i == 0 ? A.src[0] : (i == 1 ? A.src[1] : (i == 2 ? A.src[2] : A.src[3]));
}
__device__ __forceinline__ int64_t is_pick_stride(const IsArgs& A, int i) {
    return i == 0 ? A.sstride[0] : (i == 1 ? A.sstride[1] : (i == 2 ? A.sstride[2] : A.sstride[3]));
}
// (kernel-argument arrays are only ever indexed through select chains: a dynamically indexed member makes the compiler copy the whole
//  argument block into per-lane scratch -- 232 B x 2.1 M lanes = 0.5 GB of HBM writes per launch in the first build)
__device__ __forceinline__ int is_pick_wig_off(const IsArgs& A, int l) {
    return l == 0 ? A.wig_off[0] : (l == 1 ? A.wig_off[1] : (l == 2 ? A.wig_off[2] : (l == 3 ? A.wig_off[3] : (l == 4 ? A.wig_off[4] :
           (l == 5 ? A.wig_off[5] : (l == 6 ? A.wig_off[6] : A.wig_off[7]))))));
}
// LDS-DMA: per-lane global address -> LDS at (wave-uniform base + lane * size); counted by vmcnt
__device__ __forceinline__ void is_dma16(const float* __restrict__ gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
}
__device__ __forceinline__ void is_dma4(const float* __restrict__ gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 4, 0, 0);
}

// the four channels of a lane go out as one 16-byte store -- or, AT (split launches) and the segment flagged SEG_ATOMIC, as four hardware float adds
template <bool AT>
__device__ __forceinline__ void is_store4(float* __restrict__ p, f32x4 v, int flags) {
    if constexpr (AT) {
        if (flags & SEG_ATOMIC) {
#pragma unroll
            for (int k = 0; k < 4; ++k) unsafeAtomicAdd(p + k, v[k]);
            return;
        }
    }
    *reinterpret_cast<f32x4*>(p) = v;
}

// un-rotate (optional) + planar store of one segment, rows split over the four waves; D blocks staged in `dst` (see kernel)
template <int LK, int NW, bool AT = false>
__device__ __forceinline__ void epilogue_is(const IsArgs& A, const float* __restrict__ tile, const float* __restrict__ dstage, int mul_k,
                                            int out_off, int out_mulp, int flags, int64_t e, bool valid_in, int wave, int lane, const IsScan& sc) {
    bool valid = valid_in;
    constexpr int NCO = 2 * LK + 1;
    const int g = lane >> 4, el = lane & 15;
    const int rowstride = NCO * 16 + 4;
    const float* __restrict__ tl = tile + el;
    // fused scatter: the output row is the run's row, written by the slot that closes the run after the segmented scan over the 16 slots
    const bool red = A.run_id != nullptr;
    if (red) valid = sc.last && sc.row >= 0;
    float* __restrict__ ob = A.out + (red ? (int64_t)(sc.row >= 0 ? sc.row : 0) : e) * A.ostride + out_off;
    const int wend = mul_k + ((flags >> 8) & 0xff);            // + channel-padding slots of the planar block (last chunk only)
    // work unit = (output component a, 16 consecutive channels): one wave writes 64 contiguous bytes per edge and component in four
    // back-to-back stores, so the memory side sees whole sectors (interleaving the channels of one component over the waves
    // tripled the HBM write traffic: WRITE_SIZE 1.39 GB vs 0.46 GB of output per 131 072 edges)
    const int nw16 = (wend + 15) >> 4;
    const int U = NCO * nw16, per = (U + NW - 1) / NW;              // each wave takes a contiguous range of units (adjacent bytes of the row)
    const int u_begin = wave * per, u_end = (u_begin + per) < U ? (u_begin + per) : U;
    // a lane owns FOUR consecutive channels of one (edge, component): 4 NCO tile reads in flight, one 16-byte store -- a wave writes 16 edges x
    // 64 contiguous bytes per request (r3: a dword per lane, four requests for the same bytes)
    if (flags & SEG_UNROTATE) {
        const float* __restrict__ dl = dstage + el;
#pragma unroll 1
        for (int u = u_begin; u < u_end; ++u) {
            const int a = u / nw16, w = (u - a * nw16) * 16 + 4 * g;
            if (w >= wend) continue;
            float dc[NCO];
#pragma unroll
            for (int m = 0; m < NCO; ++m) dc[m] = dl[(m * NCO + a) * 16];
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* __restrict__ tw = tl + (w + k < mul_k ? w + k : 0) * rowstride;
#pragma unroll
                for (int m = 0; m < NCO; ++m) acc[k] = fmaf(dc[m], tw[m * 16], acc[k]);
                acc[k] = w + k < mul_k ? acc[k] : 0.f;
            }
            if (red) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = is_seg_scan(acc[k], sc);
            }
            if (valid) is_store4<AT>(ob + a * out_mulp + w, acc, flags);
        }
    } else {
#pragma unroll 1
        for (int u = u_begin; u < u_end; ++u) {
            const int a = u / nw16, w = (u - a * nw16) * 16 + 4 * g;
            if (w >= wend) continue;
            f32x4 acc;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = w + k < mul_k ? tl[(w + k) * rowstride + a * 16] : 0.f;
            if (red) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = is_seg_scan(acc[k], sc);
            }
            if (valid) is_store4<AT>(ob + a * out_mulp + w, acc, flags);
        }
    }
}


// A/B builds only (profiles/r06_tp_is.md section 8): the rotated staging without packed fp32 VALU instructions (K_ST_NOPK) / without 128-bit LDS writes (K_ST_W32)
__device__ __forceinline__ f32x4 st_mul(float d, f32x4 v) {
#ifdef K_ST_NOPK
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(o[k]) : "v"(d), "v"(v[k]));
    return o;
#else
    return d * v;
#endif
}
__device__ __forceinline__ void st_fma(f32x4& acc, float d, f32x4 v) {
#ifdef K_ST_PKASM                /* the packed instruction, but issued through the same kind of asm statement as K_ST_NOPK (same ordering constraints) */
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ lo = {acc[0], acc[1]}, hi = {acc[2], acc[3]};
    const f32x2_ vlo = {v[0], v[1]}, vhi = {v[2], v[3]}, dd = {d, d};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(dd), "v"(vlo));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(dd), "v"(vhi));
    acc = (f32x4){lo[0], lo[1], hi[0], hi[1]};
#elif defined(K_ST_NOPK)
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(d), "v"(v[k]));
#else
    acc += d * v;
#endif
}
__device__ __forceinline__ void st_write4(float* p, f32x4 v) {
#ifdef K_ST_W32
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<volatile float*>(p)[k] = v[k];
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
#ifndef HG_STAGE_FUSE_LMAX
#define HG_STAGE_FUSE_LMAX 6     // both node sources rotated in one pass up to this l 
#endif
// piece index t -> (component a, channel piece p): t / P1 through the reciprocal (t < 2^9, P1 <= 16: (t + 0.5) / P1 is never within 0.03 of an
// integer, so the float product rounds to the right side); an integer division per piece cost as much as the piece's own loads + FMAs at l = 0
#ifdef K_ST_IDIV                 /* A/B builds only (profiles/r06_tp_is.md section 8) */
#define HG_DIV_P1(t) ((t) / P1)
#else
#define HG_DIV_P1(t) ((int)(((float)(t) + 0.5f) * inv_P1))
#endif
#ifndef HG_STAGE_U
#define HG_STAGE_U(L) 1          // measured (profiles/r02_tp_is_experiments.md): 2-4 pieces in flight per step are SLOWER (8.39 vs 8.13 ms)
#endif
// Stage one input block (all its sources) of the workgroup's 16 edges: image offset(piece p = a * P1 + s, row e) = 64 p + 4 e per source.
//   plain source  : rows are already in the edge frame -> LDS-DMA, the four waves share the DMA instructions;
//   rotated source: rows are node features in the global frame, gathered by idx[] and multiplied by D^l(R_e) on the way in
//                   (x'[a] = sum_b D[a][b] x[b], hamgnn_amd/so3.py) -- ONCE per edge, input block and launch, which replaces the
//                   separate hg_rotate_gather pass and the materialised per-edge copies xs', xd' of the node rows.
template <int L, int NW>
__device__ __forceinline__ void stage_block(const IsArgs& A, const int* __restrict__ P, float* __restrict__ stage, int64_t erow, int wave, int lane) {
    if constexpr (NW > 4) asm volatile("" : "+v"(erow));       // 128-register budget: the rows' addresses are formed per block (hoisted out of the
                                                               // phase loop they were spilled: a scratch reload per staged piece)
    constexpr int N = 2 * L + 1;
    const int s0 = P[0], s1 = P[1], in_off = P[2], in_mulp = P[3], nsrc = P[5];
    const int g = lane >> 4, el = lane & 15;
    const int P1 = in_mulp >> 2;
    const float inv_P1 = 1.0f / (float)P1;
    const int Pfull = N * P1;
    const int nj = (Pfull + 3) >> 2;
    const bool rot0 = (A.rot_mask >> s0) & 1, rot1 = nsrc == 2 && ((A.rot_mask >> s1) & 1);
    if (rot0 && rot1 && (NW <= 4 || L <= 3) && L <= HG_STAGE_FUSE_LMAX) {
        // sender and receiver rows of the node branch share the edge's Wigner row: one output piece t = a * P1 + p (component a,
        // channels 4p..4p+3) of BOTH sources per (wave, g) slot and step -- 2 N float4 loads of the node rows + the N entries of
        // row a in flight together, 2 N float4 FMAs, two ds_write_b128 straight into the operand images
        const int64_t* __restrict__ ix0 = s0 == 0 ? A.idx[0] : (s0 == 1 ? A.idx[1] : (s0 == 2 ? A.idx[2] : A.idx[3]));
        const int64_t* __restrict__ ix1 = s1 == 0 ? A.idx[0] : (s1 == 1 ? A.idx[1] : (s1 == 2 ? A.idx[2] : A.idx[3]));
        const int64_t r0 = ix0 ? ix0[erow] : erow, r1 = ix1 ? ix1[erow] : erow;
        const float* __restrict__ row0 = is_pick_src(A, s0) + r0 * is_pick_stride(A, s0) + in_off;
        const float* __restrict__ row1 = is_pick_src(A, s1) + r1 * is_pick_stride(A, s1) + in_off;
        const float* __restrict__ D = A.wig + erow * A.nW + A.wig_off[L];
        float* __restrict__ d0 = stage + P[6] + el * 4;
        float* __restrict__ d1 = stage + P[7] + el * 4;
        // U output pieces per step (register budget by l): the node rows come from the Infinity Cache / HBM (35 MB of node rows do not
        // fit an XCD's L2), so every step of this loop exposes ~1 us of latency -- with 2 N float4 + N scalar loads of U pieces in flight
        // instead of one piece's
        constexpr int U = HG_STAGE_U(L);
#pragma unroll 1
        for (int t0 = 4 * wave + g; t0 < Pfull; t0 += 4 * NW * U) {
            f32x4 v0[U][N], v1[U][N];
            float d[U][N];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int t = t0 + 4 * NW * u;
                t = t < Pfull ? t : t0;                        // tail: re-read the first piece (result dropped below)
                const int a = HG_DIV_P1(t), p = t - a * P1;
#pragma unroll
                for (int b = 0; b < N; ++b) {
                    v0[u][b] = *reinterpret_cast<const f32x4*>(row0 + b * in_mulp + 4 * p);
                    v1[u][b] = *reinterpret_cast<const f32x4*>(row1 + b * in_mulp + 4 * p);
                    d[u][b] = D[a * N + b];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + 4 * NW * u;
                if (t < Pfull) {
                    f32x4 acc0 = st_mul(d[u][0], v0[u][0]), acc1 = st_mul(d[u][0], v1[u][0]);
#pragma unroll
                    for (int b = 1; b < N; ++b) {
                        st_fma(acc0, d[u][b], v0[u][b]);
                        st_fma(acc1, d[u][b], v1[u][b]);
                    }
                    st_write4(d0 + t * 64, acc0);
                    st_write4(d1 + t * 64, acc1);
                }
            }
        }
        return;
    }
    for (int si = 0; si < nsrc; ++si) {
        const int sidx = si ? s1 : s0;
        const int64_t* __restrict__ ix = sidx == 0 ? A.idx[0] : (sidx == 1 ? A.idx[1] : (sidx == 2 ? A.idx[2] : A.idx[3]));
        const int64_t r = ix ? ix[erow] : erow;
        const float* __restrict__ row = is_pick_src(A, sidx) + r * is_pick_stride(A, sidx) + in_off;
        float* __restrict__ dst = stage + (si ? P[7] : P[6]);
        if (si ? rot1 : rot0) {
            const float* __restrict__ D = A.wig + erow * A.nW + A.wig_off[L];
#pragma unroll 1
            for (int t = 4 * wave + g; t < Pfull; t += 4 * NW) {
                const int a = HG_DIV_P1(t), p = t - a * P1;
                f32x4 v[N];
                float d[N];
#pragma unroll
                for (int b = 0; b < N; ++b) {
                    v[b] = *reinterpret_cast<const f32x4*>(row + b * in_mulp + 4 * p);
                    d[b] = D[a * N + b];
                }
                f32x4 acc = st_mul(d[0], v[0]);
#pragma unroll
                for (int b = 1; b < N; ++b) st_fma(acc, d[b], v[b]);
                st_write4(dst + t * 64 + el * 4, acc);
            }
        } else {
#pragma unroll 1
            for (int j = wave; j < nj; j += NW) {              // the waves share the block's DMA instructions
                int p = 4 * j + g;
                p = p < Pfull ? p : Pfull - 1;
                is_dma16(row + 4 * p, dst + j * 256);
            }
        }
    }
}

#define SEG_NEWBATCH (1 << 16)

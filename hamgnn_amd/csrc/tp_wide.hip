// tp_wide.hip -- the "wide" schedule of the fused equivariant edge kernel (gfx950, CDNA4).  Hand-written HIP.  Round 5.
//
// Replaces (for large launches) the input-stationary kernel csrc/tp_is.hip, same items / fragments / weights (hamgnn_amd/plan.py), i.e. one whole
// MessagePackBlock.forward per launch (reference: hamgnn/nn/message_passing.py:191-231 incl. the node gathers and -- ConvBlock launches -- the receiver
// scatter hamgnn/nn/convolution.py:147-149).
//
// What was wrong with tp_is (four rounds of profiles, profiles/r04_tp_is_experiments.md): 56 KB of output tiles + 19 KB of staged rows per 16 edges
// => two workgroups per CU; a tile offers as many conflict-free work groups per phase as it has output segments (8) => ~8 busy waves per CU, two per
// SIMD, each waiting on L2 round trips (weights) 40 % of its time: the matrix pipe is busy half the time.  More waves per tile need finer conflict-free
// work: COLUMN WINDOWS of an item (its GEMM1 / scale / GEMM2 touch only their own columns of the tile).  The obstacle was the radial scale
// S = W3^T h of the item (27 % of all MFMAs), which every window would recompute.
//
// Here ONE workgroup of WD_NW = 16 waves (four per SIMD, <= 128 VGPRs) owns the CU's whole LDS for one 16-edge tile:
//   [ output tiles | trash row | row table | staging buffer 0 | staging buffer 1 | S buffer | S-ready flags | claim counters ]
//   * S fragments are produced ONCE per item by an "S task" into the S buffer (1 KB per 16 rows) and read by the item's column-window tasks;
//   * the staging area is double-buffered: the rows of phase p + 1 are gathered / rotated by tasks of phase p's pool;
//   * the 16 waves claim CHAINS of tasks from ONE ordered list per phase, pool(p) = [staging of p + 1 | S tasks of p | compute chains of p, largest first]
//     (a compute chain = all items of one (phase, output segment key) on one window of columns: a tile cell is updated by one wave per phase):
//     memory latency of one wave (node-row gathers, weight fragments, task records) runs under the MFMAs of the other three on its SIMD;
//     one workgroup barrier per phase.  A compute task waits for its item's S through a flag (value = pool index) -- the producer was claimed
//     earlier from the same list and never waits, so this cannot deadlock.
// Sums into a tile cell are still made by exactly one task per phase, phases are separated by barriers: the result is bit-reproducible
// and equal to tp_is's up to the order of the additions inside GEMM2's accumulator init (none: the tile value is the accumulator init there as here).
#include "tp_stage.h"

#ifndef WD_NW
#define WD_NW 16                 // waves of the workgroup = plan.WIDE_WAVES
#endif
#define WD_NT (64 * WD_NW)
#define WD_TASK_I32 32

struct WdLay {
    int sbuf_off;                // S buffer: slot s at + 256 s floats, lane's float4 at + 4 lane
    int flag_off;                // S-ready flags (ints)
    int stage_floats;            // staging buffer b at A.stage_off + b * stage_floats
    int nflag;
};

typedef volatile __attribute__((address_space(3))) int* wd_vint_p;

#define WD_MB_CASE(Q) case Q: asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "v"(x)); break;
__device__ __forceinline__ float wd_mul_bcast(float v, float x, int q) {       // x * (lane q of v's row of 16 lanes), one VALU instruction (see tp_is.hip)
    float o;
    switch (q) {
        WD_MB_CASE(0) WD_MB_CASE(1) WD_MB_CASE(2) WD_MB_CASE(3) WD_MB_CASE(4) WD_MB_CASE(5) WD_MB_CASE(6) WD_MB_CASE(7)
        WD_MB_CASE(8) WD_MB_CASE(9) WD_MB_CASE(10) WD_MB_CASE(11) WD_MB_CASE(12) WD_MB_CASE(13) WD_MB_CASE(14)
        default: asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "v"(x)); break;
    }
    return o;
}
#undef WD_MB_CASE

// ---------------------------------------------------------------------------------------------------------------- staging task
// One share (sub of nsub) of one input block of the NEXT phase: the piece loop of tp_stage.h:stage_block with (wave, NW) -> (sub, nsub).
// image offset(piece p = a * P1 + s, row e) = 64 p + 4 e per source.
template <int L>
__device__ __forceinline__ void wd_stage(const IsArgs& A, const int* __restrict__ P, float* __restrict__ stage, int64_t erow, int sub, int nsub, int lane) {
    asm volatile("" : "+v"(erow));                             // per-edge addresses are formed here, not hoisted over the task loop
    constexpr int N = 2 * L + 1;
    const int s0 = P[0], s1 = P[1], in_off = P[2], in_mulp = P[3], nsrc = P[5];
    const int g = lane >> 4, el = lane & 15;
    const int P1 = in_mulp >> 2;
    const float inv_P1 = 1.0f / (float)P1;
    const int Pfull = N * P1;
    const int nj = (Pfull + 3) >> 2;
    const bool rot0 = (A.rot_mask >> s0) & 1, rot1 = nsrc == 2 && ((A.rot_mask >> s1) & 1);
    if (rot0 && rot1 && L <= 3) {                              // both node sources share the edge's Wigner row (128-register budget: l <= 3)
        const int64_t* __restrict__ ix0 = s0 == 0 ? A.idx[0] : (s0 == 1 ? A.idx[1] : (s0 == 2 ? A.idx[2] : A.idx[3]));
        const int64_t* __restrict__ ix1 = s1 == 0 ? A.idx[0] : (s1 == 1 ? A.idx[1] : (s1 == 2 ? A.idx[2] : A.idx[3]));
        const int64_t r0 = ix0 ? ix0[erow] : erow, r1 = ix1 ? ix1[erow] : erow;
        const float* __restrict__ row0 = is_pick_src(A, s0) + r0 * is_pick_stride(A, s0) + in_off;
        const float* __restrict__ row1 = is_pick_src(A, s1) + r1 * is_pick_stride(A, s1) + in_off;
        const float* __restrict__ D = A.wig + erow * A.nW + is_pick_wig_off(A, L);
        float* __restrict__ d0 = stage + P[6] + el * 4;
        float* __restrict__ d1 = stage + P[7] + el * 4;
#pragma unroll 1
        for (int t = 4 * sub + g; t < Pfull; t += 4 * nsub) {
            const int a = HG_DIV_P1(t), p = t - a * P1;
            f32x4 v0[N], v1[N];
            float d[N];
#pragma unroll
            for (int b = 0; b < N; ++b) {
                v0[b] = *reinterpret_cast<const f32x4*>(row0 + b * in_mulp + 4 * p);
                v1[b] = *reinterpret_cast<const f32x4*>(row1 + b * in_mulp + 4 * p);
                d[b] = D[a * N + b];
            }
            f32x4 acc0 = d[0] * v0[0], acc1 = d[0] * v1[0];
#pragma unroll
            for (int b = 1; b < N; ++b) {
                acc0 += d[b] * v0[b];
                acc1 += d[b] * v1[b];
            }
            *reinterpret_cast<f32x4*>(d0 + t * 64) = acc0;
            *reinterpret_cast<f32x4*>(d1 + t * 64) = acc1;
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
            const float* __restrict__ D = A.wig + erow * A.nW + is_pick_wig_off(A, L);
#pragma unroll 1
            for (int t = 4 * sub + g; t < Pfull; t += 4 * nsub) {
                const int a = HG_DIV_P1(t), p = t - a * P1;
                f32x4 v[N];
                float d[N];
#pragma unroll
                for (int b = 0; b < N; ++b) {
                    v[b] = *reinterpret_cast<const f32x4*>(row + b * in_mulp + 4 * p);
                    d[b] = D[a * N + b];
                }
                f32x4 acc = d[0] * v[0];
#pragma unroll
                for (int b = 1; b < N; ++b) acc += d[b] * v[b];
                *reinterpret_cast<f32x4*>(dst + t * 64 + el * 4) = acc;
            }
        } else {
#pragma unroll 1
            for (int j = sub; j < nj; j += nsub) {             // plain rows (already in the edge frame): LDS-DMA, completion counted by vmcnt --
                int p = 4 * j + g;                             // waited for once, before the pool's barrier
                p = p < Pfull ? p : Pfull - 1;
                is_dma16(row + 4 * p, dst + j * 256);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- S task
// radial scale of ONE item, s_e = W3^T h2 (last layer of the radial MLP, message_passing.py:186-189 / tensor_products.py:25-47), all its RTM row
// tiles: 16 RTM MFMAs on RTM independent accumulators; the C fragments (lane (edge, g): rows 4 g + r) go to the S buffer as they are.
template <int RTM>
__device__ __forceinline__ void wd_task_S(const IsArgs& A, const WdLay& Ly, const float* __restrict__ Wb, const int* __restrict__ T, float* __restrict__ lds,
                                          int64_t erow, int lane, int stamp) {
    asm volatile("" : "+v"(erow));
    const int g = lane >> 4;
    const float* __restrict__ hrow = (T[3] ? A.h2[1] : A.h2[0]) + erow * A.hidden + 4 * g;
    const f32x4* __restrict__ w3 = reinterpret_cast<const f32x4*>(Wb + T[1]) + lane;       // [G][rt][lane]
    const int hg = __builtin_amdgcn_readfirstlane(A.hidden) >> 4;       // K groups of 16 hidden units (1..4; 4 for the shipped 64-wide layers)
    f32x4 hb[4], wv[4][RTM], S[RTM];
#pragma unroll
    for (int G = 0; G < 4; ++G)
        if (G < hg) {
            hb[G] = *reinterpret_cast<const f32x4*>(hrow + 16 * G);
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) wv[G][rt] = w3[(G * RTM + rt) * 64];
        }
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt) S[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int G = 0; G < 4; ++G)
        if (G < hg) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) S[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[G][rt][q], hb[G][q], S[rt], 0, 0, 0);
        }
    float* __restrict__ sb = lds + Ly.sbuf_off + T[4] * 256 + lane * 4;
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt) *reinterpret_cast<f32x4*>(sb + rt * 256) = S[rt];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the fragments are in the LDS before the flag is
    if (lane == 0) *(wd_vint_p)(reinterpret_cast<int*>(lds + Ly.flag_off) + T[5]) = stamp;
}

// ---------------------------------------------------------------------------------------------------------------- compute task
// NCW consecutive columns [c0, c0 + NCW) of one item: GEMM1 (A fragments x staged block) -> mid *= S * cf -> GEMM2 with the tile values as
// accumulator init (IT_TP), or GEMM1 added into the tile (IT_LIN: the PairInteractionBlock's skip o3.Linear).  Arithmetic per column exactly
// as tp_is.hip:item_is.
template <int NCW, int RTM>
__device__ __forceinline__ void wd_task_compute(const IsArgs& A, const WdLay& Ly, const float* __restrict__ Wb, const int* __restrict__ T,
                                                float* __restrict__ lds, int lane, int stamp) {
#define WD_NK2_OK(rt, r) ((rt) + 1 < RTM || 4 * (rt) + (r) < nk2)
    const int so0 = T[1], so1 = T[2], in_mulp = T[4], li = T[5], mm = T[6], neg = T[7], ksteps = T[8];
    const int c0 = T[13], x4 = T[17], nk2 = T[18], typ = T[19], rto = T[22];
    const int g = lane >> 4, el = lane & 15;
    const int* __restrict__ rtab = reinterpret_cast<const int*>(lds + A.rowtab_off) + T[23];
    float* __restrict__ tbase = lds + (el + (c0 - mm) * 16);    // + row-table entry (the row's centre column) + 16 j for window column j
    const float* __restrict__ stage = lds + A.stage_off;
    const int nsrc = so1 >= 0 ? 2 : 1;
    const int ngrp = (ksteps + 3) >> 2;
    const f32x4* __restrict__ aw = reinterpret_cast<const f32x4*>(Wb + T[11]) + lane;        // [src][G][rt][lane]
    f32x4 cfv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (typ == 0) cfv = reinterpret_cast<const f32x4*>(Wb + T[12])[lane];                    // the window's packed CG coefficients (plan.wide_schedule)
    f32x4 mid[RTM][NCW];
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
        for (int c = 0; c < NCW; ++c) mid[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int P1 = in_mulp >> 2;
    const int cdir = neg ? -P1 : P1;                           // column c -> component (neg ? a_hi - c : a_lo + c)
    const int c0p = (li - mm) * P1 + (neg ? 2 * mm * P1 : 0) + c0 * cdir;
    const int ntot = nsrc * ngrp;
    const int src_jump = (so1 - so0) - ngrp * 256;
    f32x4 av_n[RTM];
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[rt * 64];
    if (NCW <= 3 && x4) {                                      // permuted K: fragment (c, G) = piece cbase + 4 G + g of row el
        const float* __restrict__ pc[NCW];
#pragma unroll
        for (int c = 0; c < NCW; ++c) pc[c] = stage + so0 + (c0p + g + c * cdir) * 64 + el * 4;
#pragma unroll 1
        for (int t = 0; t < ntot; ++t) {
            if (t == ngrp) {
#pragma unroll
                for (int c = 0; c < NCW; ++c) pc[c] += src_jump;
            }
            f32x4 av[RTM], bv[NCW];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = av_n[rt];
            if (t + 1 < ntot) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[((t + 1) * RTM + rt) * 64];
            }
#pragma unroll
            for (int c = 0; c < NCW; ++c) {
                bv[c] = *reinterpret_cast<const f32x4*>(pc[c]);
                pc[c] += 256;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                    for (int c = 0; c < NCW; ++c)
                        mid[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][q], bv[c][q], mid[rt][c], 0, 0, 0);
        }
    } else {                                                   // natural K: element (c, 4 sl + g) = piece cbase + sl, component g
        const float* __restrict__ pc[NCW];
#pragma unroll
        for (int c = 0; c < NCW; ++c) pc[c] = stage + so0 + (c0p + c * cdir) * 64 + el * 4 + g;
        int nq = ksteps;
#pragma unroll 1
        for (int t = 0; t < ntot; ++t) {
            if (t == ngrp) {
                nq = ksteps;
#pragma unroll
                for (int c = 0; c < NCW; ++c) pc[c] += src_jump;
            }
            f32x4 av[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = av_n[rt];
            if (t + 1 < ntot) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) av_n[rt] = aw[((t + 1) * RTM + rt) * 64];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nq) {
                    float b[NCW];
#pragma unroll
                    for (int c = 0; c < NCW; ++c) b[c] = pc[c][q * 64];
                    __builtin_amdgcn_sched_barrier(0);         // operand reads together, ahead of the K-step's MFMAs (tp_is.hip, ISA audit r4)
#pragma unroll
                    for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                        for (int c = 0; c < NCW; ++c)
                            mid[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][q], b[c], mid[rt][c], 0, 0, 0);
                }
            }
            nq -= 4;
#pragma unroll
            for (int c = 0; c < NCW; ++c) pc[c] += 256;
        }
    }
    if (typ == 0) {
        const f32x4* __restrict__ a2 = reinterpret_cast<const f32x4*>(Wb + T[14]) + lane;    // [rt'][rt][lane]
        f32x4 a2_n[RTM];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[rt * 64];
        // the item's S fragments: produced by its S task, claimed earlier from this pool's list
        {
            wd_vint_p fl = (wd_vint_p)(reinterpret_cast<int*>(lds + Ly.flag_off) + T[10]);
            while (*fl != stamp) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        const float* __restrict__ sb = lds + Ly.sbuf_off + T[3] * 256 + lane * 4;
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) {
            const f32x4 S = *reinterpret_cast<const f32x4*>(sb + rt * 256);
#pragma unroll
            for (int c = 0; c < NCW; ++c) {
                const int p = rt * NCW + c;
                f32x4 t = mid[rt][c] * S;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = wd_mul_bcast(cfv[r], t[r], p);
                mid[rt][c] = t;
            }
        }
        // GEMM2: tile[w'', m] += L' fragments x mid, the tile values are the accumulator init; rows beyond mul_k go to the trash row
#pragma unroll 1
        for (int rtp = 0; rtp < rto; ++rtp) {
            f32x4 av[RTM], acc[NCW];
            float* __restrict__ trow[4];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) av[rt] = a2_n[rt];
            if (rtp + 1 < rto) {
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[((rtp + 1) * RTM + rt) * 64];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) trow[r] = tbase + rtab[16 * rtp + 4 * g + r];
#pragma unroll
            for (int c = 0; c < NCW; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[c][r] = trow[r][c * 16];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (WD_NK2_OK(rt, r)) {                    // trailing K-steps hold only padding rows: not issued
#pragma unroll
                        for (int c = 0; c < NCW; ++c)
                            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][r], mid[rt][c][r], acc[c], 0, 0, 0);
                    }
#pragma unroll
            for (int c = 0; c < NCW; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) trow[r][c * 16] = acc[c][r];
        }
    } else {
        const int row0 = T[16];
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt) {
            float* __restrict__ t0[4];
            float told[4][NCW];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t0[r] = tbase + rtab[row0 + 16 * rt + 4 * g + r];
#pragma unroll
                for (int c = 0; c < NCW; ++c) told[r][c] = t0[r][c * 16];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NCW; ++c) t0[r][c * 16] = told[r][c] + mid[rt][c][r];
        }
    }
#undef WD_NK2_OK
}

#define WD_CASE(NCWv, RTMv) case (NCWv * 8 + RTMv): wd_task_compute<NCWv, RTMv>(A, Ly, g_W, T, lds, lane, pl); break;

__global__ __launch_bounds__(WD_NT, 1) void tp_wide_kernel(const IsArgs A, const WdLay Ly, const int* __restrict__ g_segs, const int* __restrict__ g_blocks,
                                                           const int* __restrict__ g_pools, const int* __restrict__ g_chains, const int* __restrict__ g_tasks,
                                                           const float* __restrict__ g_W, const int* __restrict__ g_rowtab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = WD_NW, NT = WD_NT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + (lane & 15);
    const bool valid = e < A.rows;
    const int64_t eslot = valid ? e : A.rows - 1;
    const int64_t erow = A.eperm ? A.eperm[eslot] : eslot;      // the edge whose rows this slot reads (receiver-major launches: hamgnn_amd/topo.py)
    float* __restrict__ stage = lds + A.stage_off;

    for (int i = threadIdx.x; i < A.rowtab_off; i += NT) lds[i] = 0.f;                    // all segment tiles + the trash row
    {
        int* __restrict__ rt_l = reinterpret_cast<int*>(lds + A.rowtab_off);
        for (int i = threadIdx.x; i < A.rowtab_len; i += NT) rt_l[i] = g_rowtab[i];
        int* __restrict__ fl = reinterpret_cast<int*>(lds + Ly.flag_off);
        for (int i = threadIdx.x; i < Ly.nflag + 64; i += NT) fl[i] = 0;                  // S-ready flags, then the pools' claim counters (A.ctr_off = flag_off + nflag)
    }
    __syncthreads();
    const int npool = A.nphase + 1;
    for (int pl = 0; pl < npool; ++pl) {
        const int t0 = g_pools[2 * pl], t1 = g_pools[2 * pl + 1];
        int* __restrict__ ctr = reinterpret_cast<int*>(lds + A.ctr_off) + pl;
        while (true) {
            int ci = 0;
            if (lane == 0) ci = atomicAdd(ctr, 1);
            ci = __builtin_amdgcn_readfirstlane(ci) + t0;
            if (ci >= t1) break;
            // a chain = the records one wave runs back to back: one staging share, one S task, or all items of one (phase, output segment key)
            // restricted to a window of columns -- the tile cells of that window belong to this wave until the phase's barrier
            const int r0 = g_chains[2 * ci], r1 = g_chains[2 * ci + 1];
#pragma unroll 1
            for (int ri = r0; ri < r1; ++ri) {
                const int* __restrict__ T = g_tasks + ri * WD_TASK_I32;
                const int kind = T[0];
                if (kind == 0) {                               // a share of one input block of the next phase -> the other staging buffer
                    const int* __restrict__ B = g_blocks + T[1] * 8;
                    float* __restrict__ sbuf = stage + T[5] * Ly.stage_floats;
                    switch (T[4]) {
                        case 0: wd_stage<0>(A, B, sbuf, erow, T[2], T[3], lane); break;
                        case 1: wd_stage<1>(A, B, sbuf, erow, T[2], T[3], lane); break;
                        case 2: wd_stage<2>(A, B, sbuf, erow, T[2], T[3], lane); break;
                        case 3: wd_stage<3>(A, B, sbuf, erow, T[2], T[3], lane); break;
                        case 4: wd_stage<4>(A, B, sbuf, erow, T[2], T[3], lane); break;
                        case 5: wd_stage<5>(A, B, sbuf, erow, T[2], T[3], lane); break;
                        case 6: wd_stage<6>(A, B, sbuf, erow, T[2], T[3], lane); break;
                        default: break;
                    }
                } else if (kind == 1) {
                    switch (T[2]) {
                        case 1: wd_task_S<1>(A, Ly, g_W, T, lds, erow, lane, pl); break;
                        case 2: wd_task_S<2>(A, Ly, g_W, T, lds, erow, lane, pl); break;
                        case 3: wd_task_S<3>(A, Ly, g_W, T, lds, erow, lane, pl); break;
                        default: wd_task_S<4>(A, Ly, g_W, T, lds, erow, lane, pl); break;
                    }
                } else {
                    switch (T[15] * 8 + T[9]) {
                        WD_CASE(1, 1) WD_CASE(1, 2) WD_CASE(1, 3) WD_CASE(1, 4)
                        WD_CASE(2, 1) WD_CASE(2, 2) WD_CASE(2, 3) WD_CASE(2, 4)
                        WD_CASE(3, 1) WD_CASE(3, 2) WD_CASE(3, 3)
                        WD_CASE(4, 1) WD_CASE(4, 2)
                        WD_CASE(5, 1) WD_CASE(5, 2)
                        WD_CASE(6, 1)
                        WD_CASE(7, 1)
                        default: break;
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA of the next phase's rows has landed
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue (as tp_is.hip): all waves on one segment at a time; the Wigner
    // blocks of a batch of segments are staged together by LDS-DMA into staging buffer 0
    IsScan scan;
    scan.last = true, scan.row = 0;
    if (A.run_id) scan = is_scan_setup(valid ? A.run_id[e] : -1 - (int)(lane & 15), lane & 15);
    for (int sg = 0; sg < A.nseg; ++sg) {
        const int* __restrict__ S8 = g_segs + sg * 8;
        const int lk = S8[0], mul_k = S8[1], out_off = S8[3], out_mulp = S8[4], tile_off = S8[5], woff = S8[6], flags = S8[7];
        if (flags & SEG_NEWBATCH) {
            if (sg) __syncthreads();                           // previous batch no longer read
            int lprev = -1;
            for (int s2 = sg; s2 < A.nseg; ++s2) {
                const int* __restrict__ T8 = g_segs + s2 * 8;
                if (s2 > sg && (T8[7] & SEG_NEWBATCH)) break;
                const int l2 = T8[0];
                if (!(T8[7] & SEG_UNROTATE) || l2 == lprev) continue;
                lprev = l2;
                const int nn = (2 * l2 + 1) * (2 * l2 + 1);
                const float* __restrict__ D = A.wig + erow * A.nW + is_pick_wig_off(A, l2);
                const int nj = (nn + 3) >> 2;
#pragma unroll 1
                for (int j = wave; j < nj; j += NW) {
                    int idx = 4 * j + g;
                    idx = idx < nn ? idx : nn - 1;
                    is_dma4(D + idx, stage + T8[6] + j * 64);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const float* __restrict__ tile = lds + tile_off;
        const float* __restrict__ dst = stage + woff;
        switch (lk) {
            case 0: epilogue_is<0, NW>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 1: epilogue_is<1, NW>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 2: epilogue_is<2, NW>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 3: epilogue_is<3, NW>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 4: epilogue_is<4, NW>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 5: epilogue_is<5, NW>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            case 6: epilogue_is<6, NW>(A, tile, dst, mul_k, out_off, out_mulp, flags, e, valid, wave, lane, scan); break;
            default: break;
        }
    }
}

// lay_host, int32[12] = {nseg, nphase, trash_off, rowtab_off, rowtab_len, stage_off, stage_floats, sbuf_off, sbuf_slots, flag_off, ctr_off, lds_floats}
extern "C" int hg_tp_wide(const float* const* src, const int64_t* src_stride, int nsrc, const float* h2_node, const float* h2_edge, int hidden,
                          const float* wig, int nW, const int32_t* wig_off, const float* weights, const int32_t* seg_table, const int32_t* block_table,
                          const int32_t* pool_table, const int32_t* chain_table, const int32_t* task_table, const int32_t* row_table, const int32_t* lay_host,
                          const int64_t* const* src_idx, int rot_mask, const int64_t* edge_perm, const int32_t* run_id, float* out, int64_t out_stride,
                          int64_t rows, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nsrc < 1 || nsrc > 4) return hg_fail(-2, "hg_tp_wide: nsrc must be 1..4");
    if (hidden < 0 || hidden > 64 || (hidden & 15)) return hg_fail(-2, "hg_tp_wide: the (padded) hidden width of the radial MLP must be 0, 16, 32, 48 or 64");
    if (!lay_host || !row_table || !pool_table || !chain_table || !task_table) return hg_fail(-2, "hg_tp_wide: missing table");
    const int32_t* q = lay_host;
    const int lds_bytes = 4 * q[11];
    if (lds_bytes <= 0 || lds_bytes > 160 * 1024) return hg_fail(-2, "hg_tp_wide: bad LDS size");
    if (q[1] < 1 || q[1] + 1 > 64 || q[10] - q[9] < 1 || q[11] < q[10] + 64 || q[9] < q[7] + 256 * q[8] || q[7] < q[5] + 2 * q[6] || q[5] < q[3] + q[4] || q[3] < q[2])
        return hg_fail(-2, "hg_tp_wide: bad LDS layout");
    IsArgs A;
    for (int i = 0; i < 4; ++i) {
        A.src[i] = i < nsrc ? src[i] : src[0];
        A.sstride[i] = i < nsrc ? src_stride[i] : src_stride[0];
    }
    A.h2[0] = h2_node;
    A.h2[1] = h2_edge;
    A.hidden = hidden;
    A.wig = wig;
    A.nW = nW;
    for (int i = 0; i < 8; ++i) A.wig_off[i] = wig_off ? wig_off[i] : 0;
    A.out = out;
    A.ostride = out_stride;
    A.rows = rows;
    A.tile_shift = 0;
    A.nseg = q[0], A.nphase = q[1], A.trash_off = q[2], A.rowtab_off = q[3], A.rowtab_begin = 0, A.rowtab_len = q[4], A.stage_off = q[5], A.ctr_off = q[10];
    WdLay Ly;
    Ly.stage_floats = q[6], Ly.sbuf_off = q[7], Ly.flag_off = q[9], Ly.nflag = q[10] - q[9];
    for (int i = 0; i < 4; ++i) A.idx[i] = (src_idx && i < nsrc) ? src_idx[i] : nullptr;
    A.rot_mask = rot_mask;
    A.eperm = edge_perm;
    A.run_id = run_id;
    if (rot_mask && !wig) return hg_fail(-2, "hg_tp_wide: rotated sources need the Wigner rows");
    static unsigned char lds_attr_done[HG_MAX_DEVICES];
    if (int rc = hg_lds_attr_once(lds_attr_done, dev_guard.dev, (const void*)tp_wide_kernel, 160 * 1024)) return rc;
    const unsigned grid = (unsigned)((rows + 15) / 16);
    hipLaunchKernelGGL(tp_wide_kernel, dim3(grid), dim3(WD_NT), lds_bytes, (hipStream_t)stream, A, Ly, seg_table, block_table, pool_table, chain_table, task_table,
                       weights, row_table);
    return hg_check_launch("hg_tp_wide");
}

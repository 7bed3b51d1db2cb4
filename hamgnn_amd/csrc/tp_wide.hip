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
//   * the work of a phase -- pool(p) = S tasks of p, compute chains of p (a chain = all items of one (phase, output segment key) on one window of
//     columns: a tile cell is updated by one wave per phase), staging shares of p + 1 -- is dealt to the 16 waves by the planner (LPT on a cost model,
//     plan.wide_schedule): a wave runs ITS stream of 64-byte records back to back, so the next record is requested while the current one runs
//     (first version: dynamic claims from one list -- three dependent round trips per record, 3 k cycles of waits per ~600 cycles of MFMAs:
//     profiles/r05_tp_wide.md); memory latency of one wave runs under the MFMAs of the other three on its SIMD; one workgroup barrier per phase.
//     A compute record waits for its item's S through a flag (value = pool index): every wave runs its S tasks first and S tasks never wait, so this
//     cannot deadlock.
// Sums into a tile cell are still made by exactly one task per phase, phases are separated by barriers: the result is bit-reproducible
// and equal to tp_is's up to the order of the additions inside GEMM2's accumulator init (none: the tile value is the accumulator init there as here).
#include "tp_stage.h"

#ifndef WD_NW
#define WD_NW 16                 // waves of the workgroup = plan.WIDE_WAVES
#endif
#define WD_NT (64 * WD_NW)
#ifndef WD_A2_EARLY
#define WD_A2_EARLY 6            // accumulator fragments (row tiles x columns) up to which a record requests GEMM2's first fragments with GEMM1's
#endif
#define WD_REC_I32 16             // packed record (plan.wide_pack_record): one s_load_dwordx16
typedef int wd_rec_t __attribute__((ext_vector_type(16)));
// w0 = kind | rtm << 2 | ncw << 5 | (typ / radial generator / staging buffer) << 8 | x4 << 9 | neg << 10 | l << 11 | mm << 14 | rto << 17 | nk2 << 21 | c0 << 26
#define WD_KIND(R) ((R)[0] & 3)
#define WD_RTM(R) (((R)[0] >> 2) & 7)
#define WD_NCW(R) (((R)[0] >> 5) & 7)
#define WD_BIT8(R) (((R)[0] >> 8) & 1)
#define WD_X4(R) (((R)[0] >> 9) & 1)
#define WD_NEG(R) (((R)[0] >> 10) & 1)
#define WD_L(R) (((R)[0] >> 11) & 7)
#define WD_MM(R) (((R)[0] >> 14) & 7)
#define WD_RTO(R) (((R)[0] >> 17) & 15)
#define WD_NK2(R) (((R)[0] >> 21) & 31)
#define WD_C0(R) (((R)[0] >> 26) & 15)

// phase profiler (HG_PROF builds only, tests/bench_tp.py): per-wave shader-clock time between probes, summed over all waves
#ifdef HG_PROF
__device__ unsigned long long hg_prof_wd_acc[16];
struct ProfWd { unsigned long long t[12]; unsigned long long last; };
#ifdef HG_PROF_LITE
#define WD_PROF_ARG
#define WD_PROF_PASS
#else
#define WD_PROF_ARG , ProfWd& prof
#define WD_PROF_PASS , prof
#endif
#define WD_TPROBE(k)                                                 \
    do {                                                             \
        const unsigned long long t_ = __builtin_readcyclecounter();  \
        prof.t[k] += t_ - prof.last;                                 \
        prof.last = t_;                                              \
    } while (0)
#define WD_TL(k) WD_TPROBE(k)
#ifdef HG_PROF_LITE               // only the probes around the pools' barriers, the zero fill and the epilogue (~30 per wave and tile: no distortion)
#define WD_T(k)
#else
#define WD_T(k) WD_TPROBE(k)
#endif
#else
#define WD_PROF_ARG
#define WD_PROF_PASS
#define WD_T(k)
#define WD_TL(k)
#endif

struct WdLay {
    int sbuf_off;                // S buffer: slot s at + 256 s floats, lane's float4 at + 4 lane
    int flag_off;                // S-ready flags (ints)
    int stage_floats;            // staging buffer b at A.stage_off + b * stage_floats
    int nflag;                   // ints between flag_off and the end of the LDS image (flags + counters): zeroed at kernel start
    int own;                     // 1: "own" schedule (plan.wide_schedule(mode="own")): one record stream per wave for the whole tile, counters instead of barriers
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
__device__ __forceinline__ void wd_task_S(const IsArgs& A, const WdLay& Ly, const float* __restrict__ Wb, const wd_rec_t& T, float* __restrict__ lds,
                                          int64_t erow, int lane, int stamp) {
    asm volatile("" : "+v"(erow));
    const int g = lane >> 4;
    const float* __restrict__ hrow = (WD_BIT8(T) ? A.h2[1] : A.h2[0]) + erow * A.hidden + 4 * g;
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
    float* __restrict__ sb = lds + Ly.sbuf_off + (T[7] & 0xffff) * 256 + lane * 4;
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt) *reinterpret_cast<f32x4*>(sb + rt * 256) = S[rt];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the fragments are in the LDS before the flag is
    if (lane == 0) *(wd_vint_p)(reinterpret_cast<int*>(lds + Ly.flag_off) + (T[7] >> 16)) = stamp;
}

// "own" schedule: synchronisation records.  Monotonic counters in LDS (zeroed per tile): done[p] = waves that finished their compute records of phase p,
// staged[p] = waves whose staging shares of phase p have landed.  A wait spins (bounded: a broken schedule poisons the tile with NaN instead of hanging the GPU).
__device__ __forceinline__ void wd_sync_record(const IsArgs& A, const wd_rec_t& T, float* __restrict__ lds, int lane) {
    wd_vint_p c = (wd_vint_p)(reinterpret_cast<int*>(lds + A.ctr_off) + T[1]);
    if (T[0] & 4) {                                            // signal
        if (T[2]) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the LDS-DMA of this wave's staging shares has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads / writes so far are complete
        if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<int*>(lds + A.ctr_off) + T[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        int spin = 0;
        while (*c < T[2] && spin < (1 << 18)) {
            __builtin_amdgcn_s_sleep(1);
            ++spin;
        }
        if (spin >= (1 << 18) && lane == 0) lds[0] = __builtin_nanf("");
        asm volatile("" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------------------------- compute task
// NCW consecutive columns [c0, c0 + NCW) of one item: GEMM1 (A fragments x staged block) -> mid *= S * cf -> GEMM2 with the tile values as
// accumulator init (IT_TP), or GEMM1 added into the tile (IT_LIN: the PairInteractionBlock's skip o3.Linear).  Arithmetic per column exactly
// as tp_is.hip:item_is.
template <int NCW, int RTM>
__device__ __forceinline__ void wd_task_compute(const IsArgs& A, const WdLay& Ly, const float* __restrict__ Wb, const wd_rec_t& T,
                                                float* __restrict__ lds, int lane, int stamp WD_PROF_ARG) {
#define WD_NK2_OK(rt, r) ((rt) + 1 < RTM || 4 * (rt) + (r) < nk2)
    const int so0 = T[1], so1 = T[2], in_mulp = T[3] & 0xffff, li = WD_L(T), mm = WD_MM(T), neg = WD_NEG(T), ksteps = T[3] >> 16;
    const int c0 = WD_C0(T), x4 = WD_X4(T), nk2 = WD_NK2(T), typ = WD_BIT8(T), rto = WD_RTO(T);
    const int g = lane >> 4, el = lane & 15;
    const int* __restrict__ rtab = reinterpret_cast<const int*>(lds + A.rowtab_off) + T[9];
    float* __restrict__ tbase = lds + (el + (c0 - mm) * 16);    // + row-table entry (the row's centre column) + 16 j for window column j
    const float* __restrict__ stage = lds + A.stage_off;
    const int nsrc = so1 >= 0 ? 2 : 1;
    const int ngrp = (ksteps + 3) >> 2;
    const f32x4* __restrict__ aw = reinterpret_cast<const f32x4*>(Wb + T[4]) + lane;         // [src][G][rt][lane]
    const f32x4* __restrict__ a2 = reinterpret_cast<const f32x4*>(Wb + T[6]) + lane;         // [rt'][rt][lane]
    f32x4 cfv = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a2_n[RTM];
    // every fragment the record needs first -- GEMM1's first K group, the coefficients, GEMM2's first row tile -- is requested HERE, in one round trip
    // (GEMM2's fragments after GEMM1: a second exposed L2 latency per record; a record holds ~20 MFMAs)
    constexpr bool A2_EARLY = RTM * NCW <= WD_A2_EARLY;        // (register budget: 128; the large shapes request GEMM2's fragments after GEMM1)
    if (typ == 0) {
        cfv = reinterpret_cast<const f32x4*>(Wb + T[5])[lane];                               // the window's packed CG coefficients (plan.wide_schedule)
        if constexpr (A2_EARLY) {
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[rt * 64];
        }
    }
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
    WD_T(3);                                                    // GEMM1
    if (typ == 0) {
        if constexpr (!A2_EARLY) {
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) a2_n[rt] = a2[rt * 64];
        }
        // the item's S fragments: produced by its S task (at the head of some wave's stream of this pool)
        {
            wd_vint_p fl = (wd_vint_p)(reinterpret_cast<int*>(lds + Ly.flag_off) + (T[7] >> 16));
            int spin = 0;
            while (*fl != stamp && spin < (1 << 18)) {         // (bounded: a schedule whose S task does not precede its consumers would otherwise hang the
                __builtin_amdgcn_s_sleep(1);                   //  GPU; the bound is ~10 ms, then the rows come out as NaN instead)
                ++spin;
            }
            if (spin >= (1 << 18)) cfv = (f32x4){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
            asm volatile("" ::: "memory");
        }
        WD_T(4);                                                // waiting for the item's S fragments
        const float* __restrict__ sb = lds + Ly.sbuf_off + (T[7] & 0xffff) * 256 + lane * 4;
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
        const int row0 = T[8];
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
    WD_T(5);                                                    // scale + GEMM2 + write-back
#undef WD_NK2_OK
}

#define WD_CASE(NCWv, RTMv) case (NCWv * 8 + RTMv): wd_task_compute<NCWv, RTMv>(A, Ly, g_W, T, lds, lane, T[10] WD_PROF_PASS); break;

__global__ __launch_bounds__(WD_NT, 1) void tp_wide_kernel(const IsArgs A, const WdLay Ly, const int* __restrict__ g_segs, const int* __restrict__ g_blocks,
                                                           const int* __restrict__ g_streams, const int* __restrict__ g_recs,
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
#ifdef HG_PROF
    ProfWd prof;
    for (int k = 0; k < 12; ++k) prof.t[k] = 0;
    prof.last = __builtin_readcyclecounter();
    const unsigned long long t_begin = prof.last;
#endif

    for (int i = threadIdx.x; i < A.rowtab_off; i += NT) lds[i] = 0.f;                    // all segment tiles + the trash row
    {
        int* __restrict__ rt_l = reinterpret_cast<int*>(lds + A.rowtab_off);
        for (int i = threadIdx.x; i < A.rowtab_len; i += NT) rt_l[i] = g_rowtab[i];
        int* __restrict__ fl = reinterpret_cast<int*>(lds + Ly.flag_off);
        for (int i = threadIdx.x; i < Ly.nflag; i += NT) fl[i] = 0;                       // S-ready flags (+ the counters of the own schedule)
    }
    __syncthreads();
    WD_TL(8);                                                   // zero fill
    const int npool = A.nphase + 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // (uniform by construction; says so to the compiler: the records travel in SGPRs)
    const int npool_run = Ly.own ? 1 : npool;                   // own schedule: ONE stream per wave for the whole tile, no barrier between phases
    for (int pl = 0; pl < npool_run; ++pl) {
        // this wave's records of the pool: [S tasks | compute chains | staging shares of the next phase], back to back; the next record is requested
        // (one s_load_dwordx16) before the current one runs
        const int r0 = g_streams[2 * (pl * NW + wave_u)], r1 = g_streams[2 * (pl * NW + wave_u) + 1];
        const wd_rec_t* __restrict__ recs = reinterpret_cast<const wd_rec_t*>(g_recs);
        wd_rec_t T = recs[r0 < r1 ? r0 : 0];
#pragma unroll 1
        for (int ri = r0; ri < r1; ++ri) {
            const wd_rec_t Tn = recs[ri + 1 < r1 ? ri + 1 : ri];
            const int kind = WD_KIND(T);
            WD_T(0);                                            // record
            if (kind == 3) {
                wd_sync_record(A, T, lds, lane);
                WD_T(6);
            } else if (kind == 0) {                                   // a share of one input block of the next phase -> the other staging buffer
                const int* __restrict__ B = g_blocks + T[1] * 8;
                float* __restrict__ sbuf = stage + WD_BIT8(T) * Ly.stage_floats;
                switch (WD_L(T)) {
                    case 0: wd_stage<0>(A, B, sbuf, erow, T[2], T[3], lane); break;
                    case 1: wd_stage<1>(A, B, sbuf, erow, T[2], T[3], lane); break;
                    case 2: wd_stage<2>(A, B, sbuf, erow, T[2], T[3], lane); break;
                    case 3: wd_stage<3>(A, B, sbuf, erow, T[2], T[3], lane); break;
                    case 4: wd_stage<4>(A, B, sbuf, erow, T[2], T[3], lane); break;
                    case 5: wd_stage<5>(A, B, sbuf, erow, T[2], T[3], lane); break;
                    case 6: wd_stage<6>(A, B, sbuf, erow, T[2], T[3], lane); break;
                    default: break;
                }
                WD_T(1);                                        // staging share
            } else if (kind == 1) {
                switch (WD_RTM(T)) {
                    case 1: wd_task_S<1>(A, Ly, g_W, T, lds, erow, lane, T[10]); break;
                    case 2: wd_task_S<2>(A, Ly, g_W, T, lds, erow, lane, T[10]); break;
                    case 3: wd_task_S<3>(A, Ly, g_W, T, lds, erow, lane, T[10]); break;
                    default: wd_task_S<4>(A, Ly, g_W, T, lds, erow, lane, T[10]); break;
                }
                WD_T(2);                                        // S task
            } else {
                switch (WD_NCW(T) * 8 + WD_RTM(T)) {
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
            T = Tn;
        }
        WD_TL(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA of the next phase's rows has landed
        __syncthreads();
        WD_TL(6);                                               // waiting for the slowest wave of the pool
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
    WD_TL(7);                                                   // epilogue
#ifdef HG_PROF
    if (lane == 0) {
        for (int k = 0; k < 12; ++k) atomicAdd(&hg_prof_wd_acc[k], prof.t[k]);
        atomicAdd(&hg_prof_wd_acc[15], prof.last - t_begin);
    }
#endif
}

#ifdef HG_PROF
extern "C" int hg_prof_wd_read(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(hg_prof_wd_acc), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(hg_prof_wd_acc), z, sizeof(z));
    }
    return 0;
}
#endif

extern "C" int hg_wide_waves(void) { return WD_NW; }

// lay_host, int32[16] = {nseg, nphase, trash_off, rowtab_off, rowtab_len, stage_off, stage_floats, sbuf_off, sbuf_slots, flag_off, ctr_off, lds_floats, own, 0, 0, 0}
extern "C" int hg_tp_wide(const float* const* src, const int64_t* src_stride, int nsrc, const float* h2_node, const float* h2_edge, int hidden,
                          const float* wig, int nW, const int32_t* wig_off, const float* weights, const int32_t* seg_table, const int32_t* block_table,
                          const int32_t* stream_table, const int32_t* rec_table, const int32_t* row_table, const int32_t* lay_host,
                          const int64_t* const* src_idx, int rot_mask, const int64_t* edge_perm, const int32_t* run_id, float* out, int64_t out_stride,
                          int64_t rows, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if (nsrc < 1 || nsrc > 4) return hg_fail(-2, "hg_tp_wide: nsrc must be 1..4");
    if (hidden < 0 || hidden > 64 || (hidden & 15)) return hg_fail(-2, "hg_tp_wide: the (padded) hidden width of the radial MLP must be 0, 16, 32, 48 or 64");
    if (!lay_host || !row_table || !stream_table || !rec_table) return hg_fail(-2, "hg_tp_wide: missing table");
    const int32_t* q = lay_host;
    const int lds_bytes = 4 * q[11];
    if (lds_bytes <= 0 || lds_bytes > 160 * 1024) return hg_fail(-2, "hg_tp_wide: bad LDS size");
    if (q[1] < 1 || q[1] + 1 > 64 || q[10] - q[9] < 1 || q[11] < q[10] + (q[12] ? 128 : 1) || q[9] < q[7] + 256 * q[8] || q[7] < q[5] + 2 * q[6] || q[5] < q[3] + q[4] || q[3] < q[2])
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
    Ly.stage_floats = q[6], Ly.sbuf_off = q[7], Ly.flag_off = q[9], Ly.nflag = q[11] - q[9], Ly.own = q[12] ? 1 : 0;
    for (int i = 0; i < 4; ++i) A.idx[i] = (src_idx && i < nsrc) ? src_idx[i] : nullptr;
    A.rot_mask = rot_mask;
    A.eperm = edge_perm;
    A.run_id = run_id;
    if (rot_mask && !wig) return hg_fail(-2, "hg_tp_wide: rotated sources need the Wigner rows");
    static unsigned char lds_attr_done[HG_MAX_DEVICES];
    if (int rc = hg_lds_attr_once(lds_attr_done, dev_guard.dev, (const void*)tp_wide_kernel, 160 * 1024)) return rc;
    const unsigned grid = (unsigned)((rows + 15) / 16);
    hipLaunchKernelGGL(tp_wide_kernel, dim3(grid), dim3(WD_NT), lds_bytes, (hipStream_t)stream, A, Ly, seg_table, block_table, stream_table, rec_table,
                       weights, row_table);
    return hg_check_launch("hg_tp_wide");
}

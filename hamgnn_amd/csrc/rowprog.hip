// rowprog.hip -- a chain of ROW-LOCAL stages on feature rows held in LDS (gfx950).  Hand-written HIP.
//
// Reference: HamLayer.forward (hamgnn/models/hamgnn_output.py:51-58) = linear_transform(ResidualBlock(x)), ResidualBlock.forward
// (hamgnn/nn/interaction_blocks.py:332-358) = x + Linear2(Gate(Linear1(x))): three o3.Linears and one e3nn Gate per row, run by the
// reference (and by rounds 1-2 of this repo) as separate passes over [E, ~1000]-float rows.  Here a workgroup (8 waves) stages 16 rows once
// (contiguous 3.9 KB reads), runs every stage of plan.RowProgram LDS -> LDS and writes the last stage's rows: one read of the feature row,
// one write of the coefficient row, no intermediate leaves the chip.
//   Linear stage : block diagonal over the irreps; unit = (16 output channels of one irrep block, one input irrep) x all its components:
//                  D[channel, row] = sum_k W[k, channel] X[row, k] as v_mfma_f32_16x16x4_f32 (A = weight fragment, resident for the unit;
//                  B = one dword per lane from the source buffer, rows 4 (mod 64) floats apart: conflict-free), two accumulators in
//                  flight; the C fragment is a float4 of four consecutive channels of one row: ds_write_b128 into the destination buffer
//                  (optionally added onto what is there: the residual).  Units are dealt to the waves by cost at plan time.
//   Gate stage   : one wave per row, in place: the row's distinct activated scalars (scalars and gate channels) are overwritten by their
//                  activation, every output is then a look-up (x the gate's value); all outputs of the row are held in registers before the
//                  first is written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hg_common.h"

typedef float rp_f4 __attribute__((ext_vector_type(4)));

#ifndef RP_NW
#define RP_NW 16                // waves of the workgroup (plan.RP_NW): the stages are latency-bound chains, the LDS holds one workgroup per CU
#endif
#define RP_NT (64 * RP_NW)
#define RP_TPR (RP_NT / 16)      // threads staging one row
#define RP_TPR_SH (RP_NW == 16 ? 6 : 5)
#define RP_ROWS 16
#define RP_STAGE_I32 24
#define RP_UNIT_I32 12
#define RP_GATE_MAXV 16          // outputs of a gate row per lane (rows up to 1024 floats)

struct RpArgs {
    const float* x;
    int64_t xs;
    const int64_t* idx;          // optional row gather for the input (NULL: row r)
    float* y;
    int64_t ys;
    const float* res[2];
    int64_t rs[2];
    int64_t rows;
    int din, dout, in_buf, out_buf;
    int rsA, rsB, strip;
    int nstages;
    int nact, nout, lds_tabs;    // total entries of the gates' two tables; staged in LDS when they fit
    float cst[5];
};

__device__ __forceinline__ float rp_act(float x, int id, const float* cst) {
    switch (id) {
        case 1: return cst[1] * ((x > 15.f ? x : __logf(1.f + __expf(x))) - 0.6931471805599453f);     // shifted softplus (hardware exp / log: abs. error ~1e-7)
        case 2: return cst[2] * (1.f - __fdividef(2.f, 1.f + __expf(2.f * x)));          // tanh
        case 3: return cst[3] * __fdividef(x, 1.f + __expf(-x));
        case 4: return cst[4] * fabsf(x);
        default: return x;
    }
}

__device__ __forceinline__ rp_f4 rp_mfma(float a, float b, rp_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void rp_load_w(rp_f4 (&w)[4], const float* __restrict__ W, int w_off, int nsteps, int lane) {
    const int ngrp = (nsteps + 3) >> 2;
    const float* __restrict__ wl = W + w_off + lane * 4;
#pragma unroll
    for (int G = 0; G < 4; ++G) w[G] = G < ngrp ? *reinterpret_cast<const rp_f4*>(wl + G * 256) : rp_f4{0.f, 0.f, 0.f, 0.f};
}

// one linear unit: all components of 16 output channels of one output block from one input block; w: the unit's weight fragments.
// NG = groups of 4 K-steps (compile time: no branch between an operand read and its MFMA); the steps of the last group beyond the block's
// channels multiply zero weights -- what they read is the next component / stale row content, finite because the buffers are zeroed once
template <int NG>
__device__ __forceinline__ void rp_unit_ng(const int* __restrict__ U, const rp_f4 (&w)[4], const float* __restrict__ src, float* __restrict__ dst,
                                           int rs_src, int rs_dst, int lane) {
    const int in_off = U[0], in_mulp = U[1], out_off = U[3], out_mulp = U[4], ncomp = U[5], nv4 = U[6], acc_flag = U[8];
    const int n = lane & 15, g = lane >> 4;
    const float* __restrict__ b0 = src + n * rs_src + in_off + g;              // lane (row n, K-slot g)
    float* __restrict__ d0 = dst + n * rs_dst + out_off + 4 * g;               // lane (row n, g): channels 4 g .. 4 g + 3 of the tile
    const bool store = g < nv4;
#pragma unroll 1
    for (int m = 0; m < ncomp; m += 2) {
        const bool two = m + 1 < ncomp;
        const float* __restrict__ ba = b0 + m * in_mulp;
        const float* __restrict__ bb = two ? ba + in_mulp : ba;
        rp_f4 c0 = rp_f4{0.f, 0.f, 0.f, 0.f}, c1 = rp_f4{0.f, 0.f, 0.f, 0.f};
        if (two) {
            float xa[4 * NG], xb[4 * NG];
#pragma unroll
            for (int st = 0; st < 4 * NG; ++st) {
                xa[st] = ba[4 * st];
                xb[st] = bb[4 * st];
            }
#pragma unroll
            for (int st = 0; st < 4 * NG; ++st) {
                c0 = rp_mfma(w[st >> 2][st & 3], xa[st], c0);
                c1 = rp_mfma(w[st >> 2][st & 3], xb[st], c1);
            }
        } else {                                               // a single component: its K-steps alternate between the two accumulators
            float xa[4 * NG];
#pragma unroll
            for (int st = 0; st < 4 * NG; ++st) xa[st] = ba[4 * st];
#pragma unroll
            for (int st = 0; st < 4 * NG; st += 2) {
                c0 = rp_mfma(w[st >> 2][st & 3], xa[st], c0);
                c1 = rp_mfma(w[(st + 1) >> 2][(st + 1) & 3], xa[st + 1], c1);
            }
            c0 += c1;
        }
        if (store) {
            float* __restrict__ da = d0 + m * out_mulp;
            if (acc_flag) c0 += *reinterpret_cast<const rp_f4*>(da);
            *reinterpret_cast<rp_f4*>(da) = c0;
            if (two) {
                float* __restrict__ db = da + out_mulp;
                if (acc_flag) c1 += *reinterpret_cast<const rp_f4*>(db);
                *reinterpret_cast<rp_f4*>(db) = c1;
            }
        }
    }
}

__device__ __forceinline__ void rp_unit(const int* __restrict__ U, const rp_f4 (&w)[4], const float* __restrict__ src, float* __restrict__ dst,
                                        int rs_src, int rs_dst, int lane) {
    switch ((U[2] + 3) >> 2) {
        case 0: {                                              // an output block without a path: zeros
            const int n = lane & 15, g = lane >> 4;
            if (g < U[6] && !U[8])
                for (int m = 0; m < U[5]; ++m) *reinterpret_cast<rp_f4*>(dst + n * rs_dst + U[3] + 4 * g + m * U[4]) = rp_f4{0.f, 0.f, 0.f, 0.f};
            break;
        }
        case 1: rp_unit_ng<1>(U, w, src, dst, rs_src, rs_dst, lane); break;
        case 2: rp_unit_ng<2>(U, w, src, dst, rs_src, rs_dst, lane); break;
        case 3: rp_unit_ng<3>(U, w, src, dst, rs_src, rs_dst, lane); break;
        default: rp_unit_ng<4>(U, w, src, dst, rs_src, rs_dst, lane); break;
    }
}

#ifdef HG_PROF
__device__ unsigned long long g_rp_prof[16];
#define RP_T(k) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); g_rp_prof[k] += now_ - last_; last_ = now_; } } while (0)
extern "C" int hg_prof_rp_read(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_rp_prof), z, sizeof(z)) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rp_prof), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define RP_T(k) do { } while (0)
#endif
#define RP_PRE (256 / RP_TPR)    // float4 pieces of an input row per thread (rows up to 1024 floats)

__device__ __forceinline__ void rp_fetch_rows(const RpArgs& A, int64_t r0, rp_f4 (&pre)[RP_PRE], int tid) {
    const int row = tid >> RP_TPR_SH, j = tid & (RP_TPR - 1);
    int64_t r = r0 + row;
    r = r < A.rows ? r : A.rows - 1;
    if (A.idx) r = A.idx[r];
    const float* __restrict__ xr = A.x + r * A.xs;
    const int np = A.din >> 2;
#pragma unroll
    for (int k = 0; k < RP_PRE; ++k)
        if (j + RP_TPR * k < np) pre[k] = *reinterpret_cast<const rp_f4*>(xr + 4 * (j + RP_TPR * k));
}

extern "C" __global__ void __launch_bounds__(RP_NT)
row_program_kernel(const RpArgs A, const int* __restrict__ g_stages, const int* __restrict__ g_units, const float* __restrict__ g_W,
                   const int2* __restrict__ g_act, const int2* __restrict__ g_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* __restrict__ bufA = lds;
    float* __restrict__ bufB = lds + RP_ROWS * A.rsA;
    float* __restrict__ strips = bufB + RP_ROWS * A.rsB;
    int2* __restrict__ s_act = reinterpret_cast<int2*>(strips + RP_NW * A.strip);     // the gates' tables, staged once per (persistent) workgroup
    int2* __restrict__ s_out = s_act + A.nact;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int64_t ntiles = (A.rows + RP_ROWS - 1) / RP_ROWS;
    // this wave's first linear unit (its weight fragments are requested one unit ahead, across stages and tiles)
    int first_u = -1;
    for (int s = 0; s < A.nstages && first_u < 0; ++s) {
        const int* __restrict__ S = g_stages + s * RP_STAGE_I32;
        if (S[0] == 1 && S[4 + wave] > S[3 + wave]) first_u = S[3 + wave];
    }
    for (int i = threadIdx.x; i < RP_ROWS * (A.rsA + A.rsB); i += RP_NT) lds[i] = 0.f;      // every later read is finite (see rp_unit_ng)
    if (A.lds_tabs) {
        for (int i = threadIdx.x; i < A.nact; i += RP_NT) s_act[i] = g_act[i];
        for (int i = threadIdx.x; i < A.nout; i += RP_NT) s_out[i] = g_out[i];
    }
    __syncthreads();
    const int2* __restrict__ t_act = A.lds_tabs ? s_act : g_act;
    const int2* __restrict__ t_out = A.lds_tabs ? s_out : g_out;
    rp_f4 pre[RP_PRE];
    rp_f4 wcur[4], wnext[4];
#ifdef HG_PROF
    unsigned long long last_ = __builtin_readcyclecounter();
#endif
    rp_fetch_rows(A, (int64_t)blockIdx.x * RP_ROWS, pre, tid);
    if (first_u >= 0) rp_load_w(wcur, g_W, g_units[first_u * RP_UNIT_I32 + 7], g_units[first_u * RP_UNIT_I32 + 2], lane);
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * RP_ROWS;
        // ---- the staged rows of this tile -> LDS (32 threads per row); the next tile's rows start travelling
        {
            const int row = tid >> RP_TPR_SH, j = tid & (RP_TPR - 1);
            float* __restrict__ d = (A.in_buf ? bufB : bufA) + row * (A.in_buf ? A.rsB : A.rsA);
            const int np = A.din >> 2;
#pragma unroll
            for (int k = 0; k < RP_PRE; ++k)
                if (j + RP_TPR * k < np) *reinterpret_cast<rp_f4*>(d + 4 * (j + RP_TPR * k)) = pre[k];
        }
        __syncthreads();
        RP_T(0);
        if (tile + gridDim.x < ntiles) rp_fetch_rows(A, (tile + gridDim.x) * RP_ROWS, pre, tid);
#pragma unroll 1
        for (int s = 0; s < A.nstages; ++s) {
            const int* __restrict__ S = g_stages + s * RP_STAGE_I32;
            const int type = S[0];
            float* __restrict__ src = S[1] ? bufB : bufA;
            float* __restrict__ dst = S[2] ? bufB : bufA;
            const int rs_src = S[1] ? A.rsB : A.rsA, rs_dst = S[2] ? A.rsB : A.rsA;
            if (type == 1) {
                const int u0 = S[3 + wave], u1 = S[4 + wave];
#pragma unroll 1
                for (int u = u0; u < u1; ++u) {
                    const int* __restrict__ U = g_units + u * RP_UNIT_I32;
                    const int nu = U[9];                       // this wave's next unit (next stage / next tile included), -1: none
                    if (nu >= 0) rp_load_w(wnext, g_W, g_units[nu * RP_UNIT_I32 + 7], g_units[nu * RP_UNIT_I32 + 2], lane);
                    rp_unit(U, wcur, src, dst, rs_src, rs_dst, lane);
#pragma unroll
                    for (int G = 0; G < 4; ++G) wcur[G] = wnext[G];
                }
            } else {
                // ---- gate, in place: rows wave, wave + 8
                const int2* __restrict__ act = t_act + S[3];
                const int nact = S[4];
                const int2* __restrict__ out = t_out + S[5];
                const int Dout = S[6];
#pragma unroll 1
                for (int row = wave; row < RP_ROWS; row += RP_NW) {
                    float* __restrict__ xr = src + row * rs_src;
                    for (int i = lane; i < nact; i += 64) {    // the activated scalars overwrite their inputs
                        const int2 t = act[i];
                        xr[t.x] = rp_act(xr[t.x], t.y, A.cst);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    float v[RP_GATE_MAXV];
#pragma unroll
                    for (int k = 0; k < RP_GATE_MAXV; ++k) {
                        const int p = lane + 64 * k;
                        v[k] = 0.f;
                        if (p < Dout) {
                            const int2 t = out[p];             // {source index | -1, gate index | -1}
                            if (t.x >= 0) v[k] = t.y >= 0 ? xr[t.x] * xr[t.y] : xr[t.x];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();           // every read of the row is done before the first output overwrites it
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int k = 0; k < RP_GATE_MAXV; ++k) {
                        const int p = lane + 64 * k;
                        if (p < Dout) xr[p] = v[k];
                    }
                }
            }
            RP_T(1 + 2 * s);
            __syncthreads();
            RP_T(2 + 2 * s);
        }
        // ---- write the result rows (+ residual rows); the barrier at the top of the next tile separates these reads from its writes
        {
            const int row = tid >> RP_TPR_SH, j = tid & (RP_TPR - 1);
            const int64_t r = r0 + row;
            if (r < A.rows) {
                const float* __restrict__ sp = (A.out_buf ? bufB : bufA) + row * (A.out_buf ? A.rsB : A.rsA);
                float* __restrict__ yr = A.y + r * A.ys;
                const int np = A.dout >> 2;
                for (int p = j; p < np; p += RP_TPR) {
                    rp_f4 v = *reinterpret_cast<const rp_f4*>(sp + 4 * p);
                    if (A.res[0]) v += *reinterpret_cast<const rp_f4*>(A.res[0] + r * A.rs[0] + 4 * p);
                    if (A.res[1]) v += *reinterpret_cast<const rp_f4*>(A.res[1] + r * A.rs[1] + 4 * p);
                    *reinterpret_cast<rp_f4*>(yr + 4 * p) = v;
                }
            }
        }
        RP_T(12);
        __syncthreads();
        RP_T(13);
    }
}

// C ABI (include/hamgnn_hip.h)
extern "C" int hg_row_program(const float* x, int64_t x_stride, const int64_t* row_idx, int din, float* y, int64_t y_stride, int dout,
                              const float* res0, int64_t res0_stride, const float* res1, int64_t res1_stride,
                              const int32_t* stages, int nstages, const int32_t* units, const float* weights, const int32_t* act_tab,
                              const int32_t* out_tab, int nact, int nout, const float* consts_host, int in_buf, int out_buf, int rs_a, int rs_b, int strip,
                              int64_t rows, void* stream) {
    HgDeviceGuard dev_guard(stream);
    if (rows <= 0) return 0;
    if ((din & 3) || (dout & 3) || (rs_a & 3) || (rs_b & 3) || nstages <= 0) return hg_fail(-2, "hg_row_program: widths must be multiples of 4 floats");
    if (din > 1024) return hg_fail(-2, "hg_row_program: input rows wider than 1024 floats");
    size_t lds = sizeof(float) * ((size_t)RP_ROWS * ((size_t)rs_a + (size_t)rs_b) + (size_t)RP_NW * (size_t)strip);
    if (lds > 160 * 1024) return hg_fail(-2, "hg_row_program: the row buffers exceed the LDS");
    const size_t tabs = sizeof(int) * 2 * ((size_t)nact + (size_t)nout);
    const int lds_tabs = lds + tabs <= 160 * 1024;
    if (lds_tabs) lds += tabs;
    RpArgs A;
    A.x = x; A.xs = x_stride; A.idx = row_idx; A.y = y; A.ys = y_stride;
    A.res[0] = res0; A.rs[0] = res0_stride; A.res[1] = res1; A.rs[1] = res1_stride;
    A.rows = rows; A.din = din; A.dout = dout; A.in_buf = in_buf; A.out_buf = out_buf;
    A.rsA = rs_a; A.rsB = rs_b; A.strip = strip; A.nstages = nstages;
    A.nact = nact; A.nout = nout; A.lds_tabs = lds_tabs;
    for (int i = 0; i < 5; ++i) A.cst[i] = consts_host[i];
    static unsigned char lds_attr_done[HG_MAX_DEVICES];       // once per device (not a stream operation: illegal during graph capture)
    if (int rc = hg_lds_attr_once(lds_attr_done, dev_guard.dev, (const void*)row_program_kernel, 160 * 1024)) return rc;
    const int64_t tiles = (rows + RP_ROWS - 1) / RP_ROWS;
    const int64_t blocks = tiles < 256 ? tiles : 256;          // persistent: one workgroup per CU (the two row buffers fill its LDS), tiles strided
    row_program_kernel<<<dim3((unsigned)blocks), RP_NT, lds, (hipStream_t)stream>>>(A, stages, units, weights, (const int2*)act_tab, (const int2*)out_tab);
    return hg_check_launch("hg_row_program");
}

// pkfma_repro.hip -- SELF-CONTAINED reproducer (nothing of the library): on gfx950 (MI355X, ROCm 7.2) the packed fp32 FMA  v_pk_fma_f32 ... op_sel:[0,1,0]  (the HIGH dword of the
// src1 register pair broadcast to both lanes) returns wrong results while ANOTHER kernel issues v_mfma_f32_16x16x32_f16 on the same SIMDs.  profiles/r06_tp_is.md section 8.
//
//   hipcc --offload-arch=gfx950 -O3 -DFORM=3 tools/pkfma_repro.hip -o /tmp/pkfma_repro && /tmp/pkfma_repro [rows] [launches] [aggressor workgroups]
//       -> 77 % of the victim's rows wrong next to chains of v_mfma_f32_16x16x32_f16, none next to v_mfma_f32_16x16x16_f16 or v_mfma_f32_16x16x4_f32 chains
//   -DFORM=1 / 2 / 4 (the other three broadcast encodings), -DFORM=0 (what the compiler makes of the C loop: op_sel_hi:[0,1,1]), -DPAD=150 (victim padded to 228 VGPRs): 0 wrong rows;
//   built with -Xclang -target-feature -Xclang -packed-fp32-ops the C loop holds no packed instruction at all (what hamgnn_amd/csrc/Makefile does for the whole library).
//
// VICTIM: the edge kernel's rotated staging, reduced: per lane N float4 rows v[b] gathered from a table and N scalars d[b], acc = sum_b d[b] * v[b], written out.  No MFMA, no LDS.
// AGGRESSOR: dependent chains of one MFMA kind in registers, nothing else, on a second stream.  The victim's output is compared with its own output on an idle GPU, bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>      // 0: v_mfma_f32_16x16x32_f16   1: v_mfma_f32_16x16x16_f16   2: v_mfma_f32_16x16x4_f32
__global__ __launch_bounds__(256) void aggressor(int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f16x8 h0, hb;
    f16x4 k0, kb;
    for (int s = 0; s < 8; ++s) { h0[s] = (_Float16)(0.03f * (float)((lane + 3 * s) % 11) - 0.15f); hb[s] = (_Float16)(0.05f * (float)((lane * 3 + s) % 7) - 0.15f); }
    for (int s = 0; s < 4; ++s) { k0[s] = h0[s]; kb[s] = hb[s]; }
    const float a = 0.01f * (float)((lane * 7 + 3) % 17) - 0.08f, b = 0.02f * (float)((lane * 5 + 1) % 13) - 0.12f;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n s_nop 15\n s_nop 7" : "+v"(acc) : "v"(h0), "v"(hb));
        else if (MODE == 1)
            asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n"
                         "v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n s_nop 15\n s_nop 7" : "+v"(acc) : "v"(k0), "v"(kb));
        else
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n"
                         "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n s_nop 15\n s_nop 15\n s_nop 7" : "+v"(acc) : "v"(a), "v"(b));
        acc = acc * 0.5f;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// rows: [nrows][N][mulp] floats; dmat: [nedges][N][N]; idx: [nedges]; out: [nedges][N][mulp]   (x'[a] = sum_b D[a][b] x[b], per edge, 4 channels per lane and step)
#ifndef PAD
#define PAD 0               /* extra live registers per lane (-DPAD=176: the victim needs ~224 VGPRs like the edge kernel: two of its waves fill a SIMD's register file) */
#endif
template <int N>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ rows, const float* __restrict__ dmat, const int* __restrict__ idx, float* __restrict__ out, int nedges, int mulp) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, el = lane & 15;
    const int e = blockIdx.x * 16 + el;
    if (e >= nedges) return;
    const float* __restrict__ row = rows + (size_t)idx[e] * N * mulp;
    const float* __restrict__ D = dmat + (size_t)e * N * N;
    float* __restrict__ o = out + (size_t)e * N * mulp;
    const int P1 = mulp >> 2, Pfull = N * P1;
#if PAD > 0
    float pad[PAD];
#pragma unroll
    for (int i = 0; i < PAD; ++i) pad[i] = (float)(lane + i) * 0.001f;
#endif
#pragma unroll 1
    for (int t = 4 * wave + g; t < Pfull; t += 16) {
#if PAD > 0
#pragma unroll
        for (int i = 0; i < PAD; ++i) asm volatile("" : "+v"(pad[i]));       // all of them live across the loop body, in registers
#endif
        const int a = t / P1, p = t - a * P1;
        f32x4 v[N];
        float d[N];
#pragma unroll
        for (int b = 0; b < N; ++b) {
            v[b] = *reinterpret_cast<const f32x4*>(row + b * mulp + 4 * p);
            d[b] = D[a * N + b];
        }
#if !defined(FORM) || FORM == 0
        f32x4 acc = d[0] * v[0];
#pragma unroll
        for (int b = 1; b < N; ++b) acc += d[b] * v[b];
#else
        // the packed FMA written out, one encoding of the broadcast per FORM (the edge kernel's ISA holds all four; the compiler's choice for the loop above is form 1):
        //   1: src0 = pair (d, x), low half broadcast  op_sel_hi:[0,1,1]      2: src1 = pair (d, x), low half broadcast  op_sel_hi:[1,0,1]
        //   3: src1 = pair (x, d), HIGH half broadcast op_sel:[0,1,0]         4: src0 = pair (x, d), HIGH half broadcast op_sel:[1,0,0]
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        f32x2_ lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < N; ++b) {
            const float other = d[(b + 1) % N];
            const f32x2_ vlo = {v[b][0], v[b][1]}, vhi = {v[b][2], v[b][3]};
#if FORM == 1
            const f32x2_ dp = {d[b], other};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(lo) : "v"(dp), "v"(vlo));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(hi) : "v"(dp), "v"(vhi));
#elif FORM == 2
            const f32x2_ dp = {d[b], other};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(lo) : "v"(vlo), "v"(dp));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(hi) : "v"(vhi), "v"(dp));
#elif FORM == 3
            const f32x2_ dp = {other, d[b]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(lo) : "v"(vlo), "v"(dp));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(hi) : "v"(vhi), "v"(dp));
#else
            const f32x2_ dp = {other, d[b]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(lo) : "v"(dp), "v"(vlo));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(hi) : "v"(dp), "v"(vhi));
#endif
        }
        const f32x4 acc = {lo[0], lo[1], hi[0], hi[1]};
#endif
        *reinterpret_cast<f32x4*>(o + a * mulp + 4 * p) = acc;
    }
#if PAD > 0
    float sp = 0.f;
#pragma unroll
    for (int i = 0; i < PAD; ++i) sp += pad[i];
    if (sp == 12345.678f) o[0] = sp;                             // (never true: keeps the registers' values needed)
#endif
}

int main(int argc, char** argv) {
    const int nedges = argc > 1 ? atoi(argv[1]) : 262144, reps = argc > 2 ? atoi(argv[2]) : 5, ag_grid = argc > 3 ? atoi(argv[3]) : 256;
    constexpr int N = 7;                                          // l = 3
    const int mulp = 32, nrows = 16384;
    std::vector<float> hrows((size_t)nrows * N * mulp), hd((size_t)nedges * N * N);
    std::vector<int> hidx(nedges);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& x : hrows) x = rnd();
    for (auto& x : hd) x = rnd();
    for (auto& x : hidx) { s = s * 1664525u + 1013904223u; x = (int)((s >> 8) % nrows); }
    float *rows, *dm, *out, *ag_out;
    int* idx;
    const size_t nout = (size_t)nedges * N * mulp;
    CHECK(hipMalloc(&rows, hrows.size() * 4)); CHECK(hipMalloc(&dm, hd.size() * 4)); CHECK(hipMalloc(&idx, hidx.size() * 4)); CHECK(hipMalloc(&out, nout * 4));
    CHECK(hipMalloc(&ag_out, (size_t)ag_grid * 256 * 4));
    CHECK(hipMemcpy(rows, hrows.data(), hrows.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dm, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sv;
    CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sv));
    const int vgrid = (nedges + 15) / 16;
    std::vector<float> ref(nout), got(nout);
    hipLaunchKernelGGL(victim<N>, dim3(vgrid), dim3(256), 0, sv, rows, dm, idx, out, nedges, mulp);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ref.data(), out, nout * 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, sv)); hipLaunchKernelGGL(victim<N>, dim3(vgrid), dim3(256), 0, sv, rows, dm, idx, out, nedges, mulp); CHECK(hipEventRecord(e1, sv)); CHECK(hipDeviceSynchronize());
    float v_ms = 0.f; CHECK(hipEventElapsedTime(&v_ms, e0, e1));
    const char* names[3] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x4_f32"};
#ifdef __HIP_DEVICE_COMPILE__
#endif
    for (int mode : {2, 1, 0}) {
        auto launch_ag = [&](int iters) {
            if (mode == 0) hipLaunchKernelGGL(aggressor<0>, dim3(ag_grid), dim3(256), 0, sa, iters, ag_out);
            else if (mode == 1) hipLaunchKernelGGL(aggressor<1>, dim3(ag_grid), dim3(256), 0, sa, iters, ag_out);
            else hipLaunchKernelGGL(aggressor<2>, dim3(ag_grid), dim3(256), 0, sa, iters, ag_out);
        };
        CHECK(hipEventRecord(e0, sa)); launch_ag(20000); CHECK(hipEventRecord(e1, sa)); CHECK(hipDeviceSynchronize());
        float a_ms = 0.f; CHECK(hipEventElapsedTime(&a_ms, e0, e1));
        const int iters = (int)(20000.0 * (6.0 * v_ms + 20.0) / (a_ms > 1e-3f ? a_ms : 1e-3f)) + 1000;
        long bad_elems = 0, bad_edges = 0, bad_runs = 0;
        double worst = 0.0;
        for (int rep = 0; rep < reps; ++rep) {
            CHECK(hipMemset(out, 0, nout * 4));
            CHECK(hipDeviceSynchronize());
            launch_ag(iters);
            hipLaunchKernelGGL(victim<N>, dim3(vgrid), dim3(256), 0, sv, rows, dm, idx, out, nedges, mulp);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(got.data(), out, nout * 4, hipMemcpyDeviceToHost));
            long be = 0;
            for (int e = 0; e < nedges; ++e) {
                const size_t o = (size_t)e * N * mulp;
                if (memcmp(&got[o], &ref[o], (size_t)N * mulp * 4)) {
                    ++be;
                    for (size_t i = o; i < o + (size_t)N * mulp; ++i)
                        if (got[i] != ref[i]) { ++bad_elems; const double r = fabs((double)got[i] - ref[i]) / (fabs((double)ref[i]) + 1e-6); if (r > worst) worst = r; }
                }
            }
            bad_edges += be;
            bad_runs += be > 0;
        }
        printf("{\"victim\": \"acc += d * v on float4s (the compiler's packed fp32 form unless built with -packed-fp32-ops)\", \"aggressor\": \"chains of %s\", \"launches\": %d, \"launches_with_wrong_rows\": %ld, "
               "\"wrong_rows\": %ld, \"of_rows\": %ld, \"wrong_elements\": %ld, \"worst_rel\": %.3e, \"victim_ms_alone\": %.3f}\n",
               names[mode], reps, bad_runs, bad_edges, (long)nedges * reps, bad_elems, worst, v_ms);
        fflush(stdout);
    }
    return 0;
}

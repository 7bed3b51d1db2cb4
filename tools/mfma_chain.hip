// mfma_chain.hip -- standalone probe (no part of the library): BACK-TO-BACK DEPENDENT MFMA chains (the accumulator of one MFMA is the C operand of the very next
// instruction -- the edge kernel has 460 such pairs of v_mfma_f32_16x16x4_f32 and 78 of v_mfma_f32_16x16x32_f16) while the partner wave of the SIMD issues MFMAs of the
// other kind.  tools/mfma_mix.hip only had chains at distance 4.  Background: profiles/r06_tp_is.md section 4.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain [workgroups] [iterations] [launches]
//
// A wave's result is a pure function of (mode, wave, workgroup parity); the grid runs with 160 KB of LDS per workgroup (one wave per SIMD: the reference) and with 80 KB
// (two waves per SIMD); results must agree bit for bit.  The A operand changes from MFMA to MFMA inside a chain, so a C operand read too early loses a term.
//   mode 0  fp32 chains only          mode 1  half-precision chains only        mode 2  every wave alternates chains of both kinds (lengths drift per wave / parity)
//   mode 3  even workgroups fp32 chains, odd workgroups half-precision chains
//   mode 4  inside one wave: fp32 and half-precision MFMAs alternate instruction by instruction, each kind chained on its own accumulator
//   mode 5  as 2, and the chains' accumulators travel through the LDS between chains (ds_write right behind the last MFMA of a chain -- compiler-scheduled)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// eight dependent MFMAs, nothing between them; four different A operands
__device__ __forceinline__ void chain_f32(f32x4& acc, float a0, float a1, float a2, float a3, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %2, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %3, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n"
                 "v_mfma_f32_16x16x4_f32 %0, %2, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %3, %5, %0\n"
                 "s_nop 15\n s_nop 15\n s_nop 7"
                 : "+v"(acc) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));
}
__device__ __forceinline__ void chain_f16(f32x4& acc, f16x8 a0, f16x8 a1, f16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n"
                 "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n"
                 "s_nop 15\n s_nop 7"
                 : "+v"(acc) : "v"(a0), "v"(a1), "v"(b));
}
// compiler-scheduled (the hazard recogniser inserts what the ISA asks for between MFMAs of different kinds)
__device__ __forceinline__ void chain_alt(f32x4& accF, f32x4& accH, float a0, float a1, float b, f16x8 h0, f16x8 h1, f16x8 hb) {
    accF = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, accF, 0, 0, 0);
    accH = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, hb, accH, 0, 0, 0);
    accF = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, accF, 0, 0, 0);
    accH = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, hb, accH, 0, 0, 0);
    accF = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, accF, 0, 0, 0);
    accH = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, hb, accH, 0, 0, 0);
    accF = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, accF, 0, 0, 0);
    accH = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, hb, accH, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void chain_kernel(int mode, int iters, float* __restrict__ out) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) lds[0] = 0.f;
    f32x4 accF = (f32x4){0.f, 0.f, 0.f, 0.f}, accH = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a0 = 0.01f * (float)((lane * 7 + 3) % 17) - 0.08f, a1 = 0.013f * (float)((lane * 3 + 5) % 19) - 0.11f, a2 = 0.017f * (float)((lane * 11 + 1) % 13) - 0.1f,
                a3 = 0.007f * (float)((lane * 5 + 7) % 23) - 0.07f, b = 0.02f * (float)((lane * 5 + 1) % 13) - 0.12f;
    f16x8 h0, h1, hb;
    for (int s = 0; s < 8; ++s) {
        h0[s] = (_Float16)(0.03f * (float)((lane + 3 * s) % 11) - 0.15f);
        h1[s] = (_Float16)(0.02f * (float)((lane * 5 + s) % 13) - 0.12f);
        hb[s] = (_Float16)(0.05f * (float)((lane * 3 + s) % 7) - 0.15f);
    }
    const int par = blockIdx.x & 1;
    const int nf = 2 + wave + par, nh = 4 - wave + 2 * par;      // chains per round: the two waves of a SIMD drift against each other
    float* tile = lds + 64 + wave * 512;
    if (mode == 5) {
        for (int r = 0; r < 4; ++r) { tile[r * 64 + lane] = 0.f; tile[256 + r * 64 + lane] = 0.f; }
    }
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (mode == 0 || (mode == 3 && !par)) {
            for (int k = 0; k < nf; ++k) { chain_f32(accF, a0, a1, a2, a3, b); accF = accF * 0.5f; }
        } else if (mode == 1 || (mode == 3 && par)) {
            for (int k = 0; k < nh; ++k) { chain_f16(accH, h0, h1, hb); accH = accH * 0.5f; }
        } else if (mode == 2) {
            for (int k = 0; k < nf; ++k) { chain_f32(accF, a0, a1, a2, a3, b); accF = accF * 0.5f; }
            for (int k = 0; k < nh; ++k) { chain_f16(accH, h0, h1, hb); accH = accH * 0.5f; }
        } else if (mode == 4) {
            for (int k = 0; k < nf; ++k) { chain_alt(accF, accH, a0, a1, b, h0, h1, hb); accF = accF * 0.5f; accH = accH * 0.5f; }
        } else {
            for (int k = 0; k < nf; ++k) {
                f32x4 t;
                for (int r = 0; r < 4; ++r) t[r] = tile[r * 64 + lane] * 0.5f;
                t = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, t, 0, 0, 0);
                for (int r = 0; r < 4; ++r) tile[r * 64 + lane] = t[r];
            }
            for (int k = 0; k < nh; ++k) {
                f32x4 t;
                for (int r = 0; r < 4; ++r) t[r] = tile[256 + r * 64 + lane] * 0.5f;
                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, hb, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, hb, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, hb, t, 0, 0, 0);
                for (int r = 0; r < 4; ++r) tile[256 + r * 64 + lane] = t[r];
            }
        }
    }
    if (mode == 5) {
        for (int r = 0; r < 4; ++r) { accF[r] = tile[r * 64 + lane]; accH[r] = tile[256 + r * 64 + lane]; }
    }
    float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    for (int r = 0; r < 4; ++r) { o[r] = accF[r]; o[4 + r] = accH[r]; }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 1500, reps = argc > 3 ? atoi(argv[3]) : 10;
    const size_t n = (size_t)grid * 256 * 8;
    float* d;
    CHECK(hipMalloc(&d, n * sizeof(float)));
    CHECK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<float> ref(n), got(n);
    const char* names[6] = {"fp32 chains only", "half-precision chains only", "every wave alternates chains of both kinds", "even workgroups fp32 chains / odd half-precision chains",
                            "fp32 and half-precision MFMAs alternate instruction by instruction", "as 2, accumulators travel through the LDS between chains"};
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(256), 160 * 1024, 0, mode, iters, d);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(ref.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        double norm = 0.0;
        for (size_t i = 0; i < n; ++i) norm += fabs((double)ref[i]);
        for (int lds_kb : {160, 80}) {
            long bad_runs = 0, bad_wgs = 0;
            double worst = 0.0;
            for (int rep = 0; rep < reps; ++rep) {
                CHECK(hipMemset(d, 0, n * sizeof(float)));
                hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(256), lds_kb * 1024, 0, mode, iters, d);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(got.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
                long bw = 0;
                for (int wg = 0; wg < grid; ++wg) {
                    const size_t o = (size_t)wg * 256 * 8;
                    if (memcmp(&got[o], &ref[o], 256 * 8 * sizeof(float))) {
                        ++bw;
                        for (size_t i = o; i < o + 256 * 8; ++i) { double e = fabs((double)got[i] - ref[i]) / (fabs((double)ref[i]) + 1e-30); if (e > worst && ref[i] != 0.f) worst = e; }
                    }
                }
                bad_wgs += bw;
                bad_runs += bw > 0;
            }
            printf("{\"mode\": %d, \"what\": \"%s\", \"waves_per_simd\": %d, \"launches\": %d, \"launches_with_wrong_workgroups\": %ld, \"wrong_workgroups\": %ld, \"of\": %ld, \"worst_rel\": %.3e, \"mean_abs_ref\": %.3e}\n",
                   mode, names[mode], lds_kb == 160 ? 1 : 2, reps, bad_runs, bad_wgs, (long)grid * reps, worst, norm / (double)n);
            fflush(stdout);
        }
    }
    CHECK(hipFree(d));
    return 0;
}

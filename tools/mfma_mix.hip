// mfma_mix.hip -- standalone probe (no part of the library): do half-precision MFMAs (v_mfma_f32_16x16x32_f16) and fp32-input MFMAs (v_mfma_f32_16x16x4_f32) of
// DIFFERENT waves on one SIMD disturb each other on gfx950?  Background: profiles/r06_tp_is.md section 4 -- the edge kernel with its radial scales on the half-precision pipe
// produced rare wrong rows only when a second wave shared the SIMD, although every scale checked out in-kernel.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
//
// Every wave runs a fixed, data-independent stream of MFMA bursts on constant operands (accumulators damped by a VALU multiply so they stay finite) and writes its
// accumulators; a wave's result is a pure function of (mode, wave, workgroup parity).  The SAME grid is run with 160 KB of LDS per workgroup (one workgroup = one wave per
// SIMD: the reference) and with 80 KB (two workgroups per CU = two waves per SIMD); results must agree bit for bit.  Modes: 0 all fp32-input MFMAs, 1 all half-precision
// MFMAs, 2 even workgroups fp32 / odd workgroups half precision, 3 every wave alternates bursts of both kinds with lengths that differ per wave (desynchronised).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void burst_f32(f32x4 (&acc)[4], float a, float b, int n) {
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + 0.125f * k, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = acc[k] * 0.75f;
    }
}
__device__ __forceinline__ void burst_f16(f32x4 (&acc)[4], f16x8 a, f16x8 b, int n) {
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = acc[k] * 0.75f;
    }
}

// mode 4: the pattern of the edge kernel's GEMM2 -- accumulators initialised FROM the LDS, one short chain of fp32-input MFMAs, results stored straight back to the LDS
// (the compiler's wait states between the MFMA and the ds_write are all that separates them) -- interleaved with bursts of half-precision MFMAs, per wave
__device__ __forceinline__ void rmw_f32(float* __restrict__ tile, float a, float b, int n, int lane) {
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[k][r] = tile[(k * 4 + r) * 64 + lane] * 0.75f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + 0.125f * k, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(k * 4 + r) * 64 + lane] = acc[k][r];
    }
}

__global__ __launch_bounds__(256, 2) void mix_kernel(int mode, int iters, float* __restrict__ out) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) lds[0] = 0.f;                          // (the dynamic LDS is what sets the workgroups per CU)
    f32x4 accF[4], accH[4];
    for (int k = 0; k < 4; ++k) { accF[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; accH[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float a = 0.01f * (float)((lane * 7 + 3) % 17) - 0.08f, b = 0.02f * (float)((lane * 5 + 1) % 13) - 0.12f;
    f16x8 a8, b8;
    for (int s = 0; s < 8; ++s) { a8[s] = (_Float16)(0.03f * (float)((lane + 3 * s) % 11) - 0.15f); b8[s] = (_Float16)(0.05f * (float)((lane * 3 + s) % 7) - 0.15f); }
    const int par = blockIdx.x & 1;
    if (mode == 0) burst_f32(accF, a, b, iters);
    else if (mode == 1) burst_f16(accH, a8, b8, iters);
    else if (mode == 4) {
        float* tile = lds + 64 + wave * 1024;
        for (int i = lane; i < 1024; i += 64) tile[i] = 0.f;
        const int nf = 2 + wave + par, nh = 4 - wave + 2 * par;
        for (int r = 0; r < iters / 8; ++r) {
            rmw_f32(tile, a, b, nf, lane);
            burst_f16(accH, a8, b8, nh);
        }
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < 4; ++r) accF[k][r] = tile[(k * 4 + r) * 64 + lane];
    }
    else if (mode == 2) { if (par) burst_f16(accH, a8, b8, iters); else burst_f32(accF, a, b, iters / 2); }
    else {
        const int nf = 3 + wave + 2 * par, nh = 5 - wave + par;  // burst lengths differ per wave and workgroup parity: the two waves of a SIMD drift against each other
        for (int r = 0; r < iters / 8; ++r) {
            burst_f32(accF, a, b, nf);
            burst_f16(accH, a8, b8, nh);
        }
    }
    float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 32;
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 4; ++r) { o[k * 4 + r] = accF[k][r]; o[16 + k * 4 + r] = accH[k][r]; }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 4000, reps = argc > 3 ? atoi(argv[3]) : 20;
    const size_t n = (size_t)grid * 256 * 32;
    float* d;
    CHECK(hipMalloc(&d, n * sizeof(float)));
    CHECK(hipFuncSetAttribute((const void*)mix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<float> ref(n), got(n);
    const char* names[5] = {"all fp32-input MFMAs", "all half-precision MFMAs", "even workgroups fp32 / odd half precision", "every wave alternates bursts of both kinds",
                            "fp32 MFMA chains that read-modify-write an LDS tile, interleaved with half-precision bursts"};
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(256), 160 * 1024, 0, mode, iters, d);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(ref.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        for (int lds_kb : {160, 80}) {
            long bad_runs = 0, bad_wgs = 0;
            double worst = 0.0;
            for (int rep = 0; rep < reps; ++rep) {
                CHECK(hipMemset(d, 0, n * sizeof(float)));
                hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(256), lds_kb * 1024, 0, mode, iters, d);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(got.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
                long bw = 0;
                for (int wg = 0; wg < grid; ++wg) {
                    const size_t o = (size_t)wg * 256 * 32;
                    if (memcmp(&got[o], &ref[o], 256 * 32 * sizeof(float))) {
                        ++bw;
                        for (size_t i = o; i < o + 256 * 32; ++i) { double e = fabs((double)got[i] - ref[i]) / (fabs((double)ref[i]) + 1e-30); if (e > worst && ref[i] != 0.f) worst = e; }
                    }
                }
                bad_wgs += bw;
                bad_runs += bw > 0;
            }
            printf("{\"mode\": %d, \"what\": \"%s\", \"lds_kb_per_workgroup\": %d, \"waves_per_simd\": %d, \"launches\": %d, \"launches_with_wrong_workgroups\": %ld, \"wrong_workgroups\": %ld, \"of\": %ld, \"worst_rel\": %.3e}\n",
                   mode, names[mode], lds_kb, lds_kb == 160 ? 1 : 2, reps, bad_runs, bad_wgs, (long)grid * reps, worst);
        }
    }
    CHECK(hipFree(d));
    return 0;
}

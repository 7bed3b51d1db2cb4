// mfma_pair.hip -- standalone probe (no part of the library): two DIFFERENT MFMA streams on the two waves that share a SIMD, gfx950.
// Background: profiles/r06_tp_is.md section 8 -- the shipped fp32 edge kernel computes wrong tiles as soon as DEPENDENT chains of v_mfma_f32_16x16x32_f16 (results unused,
// operands unrelated) are added to it; independent ones, a single one, or the K = 16 form are harmless.  tools/mfma_mix.hip / mfma_chain.hip gave the two roles to even / odd
// workgroups -- but workgroups are dealt round-robin over the 8 XCDs, so the two workgroups that share a CU always had the SAME parity and the roles never met on a SIMD.
// Here the role is bit 8 of the workgroup index (workgroups j and j + 256 are the ones that share a CU when 512 are resident).
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_pair.hip -o /tmp/mfma_pair && /tmp/mfma_pair [workgroups] [iterations] [launches]
//
// Roles: F = back-to-back dependent chains of 8 v_mfma_f32_16x16x4_f32;  H = dependent chains of 6 v_mfma_f32_16x16x32_f16;  I = 6 INDEPENDENT v_mfma_f32_16x16x32_f16;
// K = dependent chains of 6 v_mfma_f32_16x16x16_f16;  L = F with its accumulator travelling through the LDS between chains.  Every wave's result is a pure function of
// (role, wave); reference = the same grid with 160 KB of LDS per workgroup (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

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
__device__ __forceinline__ void indep_f16(f32x4 (&acc)[6], f16x8 a0, f16x8 a1, f16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %6, %8, %0\n v_mfma_f32_16x16x32_f16 %1, %7, %8, %1\n v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n v_mfma_f32_16x16x32_f16 %3, %7, %8, %3\n"
                 "v_mfma_f32_16x16x32_f16 %4, %7, %8, %4\n v_mfma_f32_16x16x32_f16 %5, %6, %8, %5\n"
                 "s_nop 15\n s_nop 7"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]) : "v"(a0), "v"(a1), "v"(b));
}
__device__ __forceinline__ void chain_k16(f32x4& acc, f16x4 a0, f16x4 a1, f16x4 b) {
    asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %3, %0\n v_mfma_f32_16x16x16_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %3, %0\n v_mfma_f32_16x16x16_f16 %0, %2, %3, %0\n"
                 "v_mfma_f32_16x16x16_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %3, %0\n"
                 "s_nop 15\n s_nop 7"
                 : "+v"(acc) : "v"(a0), "v"(a1), "v"(b));
}

// role codes
enum { R_F = 0, R_H = 1, R_I = 2, R_K = 3, R_L = 4 };

__global__ __launch_bounds__(256, 2) void pair_kernel(int role0, int role1, int role_bit, int iters, float* __restrict__ out) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) lds[0] = 0.f;
    const int role = ((blockIdx.x >> role_bit) & 1) ? role1 : role0;
    f32x4 acc[6];
    for (int k = 0; k < 6; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a0 = 0.01f * (float)((lane * 7 + 3) % 17) - 0.08f, a1 = 0.013f * (float)((lane * 3 + 5) % 19) - 0.11f, a2 = 0.017f * (float)((lane * 11 + 1) % 13) - 0.1f,
                a3 = 0.007f * (float)((lane * 5 + 7) % 23) - 0.07f, b = 0.02f * (float)((lane * 5 + 1) % 13) - 0.12f;
    f16x8 h0, h1, hb;
    f16x4 k0, k1, kb;
    for (int s = 0; s < 8; ++s) {
        h0[s] = (_Float16)(0.03f * (float)((lane + 3 * s) % 11) - 0.15f);
        h1[s] = (_Float16)(0.02f * (float)((lane * 5 + s) % 13) - 0.12f);
        hb[s] = (_Float16)(0.05f * (float)((lane * 3 + s) % 7) - 0.15f);
    }
    for (int s = 0; s < 4; ++s) { k0[s] = h0[s]; k1[s] = h1[s]; kb[s] = hb[s]; }
    float* tile = lds + 64 + wave * 256;
    if (role == R_L)
        for (int r = 0; r < 4; ++r) tile[r * 64 + lane] = 0.f;
    const int n = 3 + wave;                                      // chains per round (per wave: the four SIMDs see different rhythms)
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (role == R_F) {
            for (int k = 0; k < n; ++k) { chain_f32(acc[0], a0, a1, a2, a3, b); acc[0] = acc[0] * 0.5f; }
        } else if (role == R_H) {
            for (int k = 0; k < n; ++k) { chain_f16(acc[0], h0, h1, hb); acc[0] = acc[0] * 0.5f; }
        } else if (role == R_I) {
            for (int k = 0; k < n; ++k) {
                indep_f16(acc, h0, h1, hb);
                for (int j = 0; j < 6; ++j) acc[j] = acc[j] * 0.5f;
            }
        } else if (role == R_K) {
            for (int k = 0; k < n; ++k) { chain_k16(acc[0], k0, k1, kb); acc[0] = acc[0] * 0.5f; }
        } else {
            for (int k = 0; k < n; ++k) {
                f32x4 t;
                for (int r = 0; r < 4; ++r) t[r] = tile[r * 64 + lane] * 0.5f;
                chain_f32(t, a0, a1, a2, a3, b);
                for (int r = 0; r < 4; ++r) tile[r * 64 + lane] = t[r];
            }
        }
    }
    if (role == R_L)
        for (int r = 0; r < 4; ++r) acc[0][r] = tile[r * 64 + lane];
    float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    for (int r = 0; r < 4; ++r) { o[r] = acc[0][r]; o[4 + r] = acc[1][r] + acc[2][r] + acc[3][r] + acc[4][r] + acc[5][r]; }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 1500, reps = argc > 3 ? atoi(argv[3]) : 10;
    const size_t n = (size_t)grid * 256 * 8;
    float* d;
    CHECK(hipMalloc(&d, n * sizeof(float)));
    CHECK(hipFuncSetAttribute((const void*)pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<float> ref(n), got(n);
    const char* rn[5] = {"F: dependent fp32 chains", "H: dependent 16x16x32 f16 chains", "I: independent 16x16x32 f16", "K: dependent 16x16x16 f16 chains", "L: fp32 chains through the LDS"};
    const int pairs[][2] = {{R_F, R_H}, {R_F, R_I}, {R_F, R_K}, {R_L, R_H}, {R_F, R_F}, {R_H, R_H}, {R_H, R_I}};
    for (int role_bit : {8, 0})
        for (auto& pr : pairs) {
            hipLaunchKernelGGL(pair_kernel, dim3(grid), dim3(256), 160 * 1024, 0, pr[0], pr[1], role_bit, iters, d);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(ref.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
            for (int lds_kb : {160, 80}) {
                long bad_runs = 0, bad[2] = {0, 0};
                double worst = 0.0;
                for (int rep = 0; rep < reps; ++rep) {
                    CHECK(hipMemset(d, 0, n * sizeof(float)));
                    hipLaunchKernelGGL(pair_kernel, dim3(grid), dim3(256), lds_kb * 1024, 0, pr[0], pr[1], role_bit, iters, d);
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipMemcpy(got.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
                    long bw = 0;
                    for (int wg = 0; wg < grid; ++wg) {
                        const size_t o = (size_t)wg * 256 * 8;
                        if (memcmp(&got[o], &ref[o], 256 * 8 * sizeof(float))) {
                            ++bw;
                            ++bad[(wg >> role_bit) & 1];
                            for (size_t i = o; i < o + 256 * 8; ++i) { double e = fabs((double)got[i] - ref[i]) / (fabs((double)ref[i]) + 1e-30); if (e > worst && ref[i] != 0.f) worst = e; }
                        }
                    }
                    bad_runs += bw > 0;
                }
                printf("{\"role_bit\": %d, \"role0\": \"%s\", \"role1\": \"%s\", \"waves_per_simd\": %d, \"launches\": %d, \"launches_with_wrong_workgroups\": %ld, \"wrong_workgroups_role0\": %ld, \"wrong_workgroups_role1\": %ld, \"of\": %ld, \"worst_rel\": %.3e}\n",
                       role_bit, rn[pr[0]], rn[pr[1]], lds_kb == 160 ? 1 : 2, reps, bad_runs, bad[0], bad[1], (long)grid * reps, worst);
                fflush(stdout);
            }
        }
    CHECK(hipFree(d));
    return 0;
}

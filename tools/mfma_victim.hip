// mfma_victim.hip -- standalone two-kernel reproducer (no part of the library), gfx950: an AGGRESSOR kernel that only issues MFMAs in registers (no LDS, no memory traffic)
// runs on one stream while a VICTIM kernel of plain fp32 work runs on another; the victim's result is compared with its own result on an idle GPU.
// Background: profiles/r06_tp_is.md section 8 -- the unmodified shipped edge kernel (fp32 MFMAs only) computes wrong tiles while a separate kernel issues
// v_mfma_f32_16x16x32_f16 / _bf16 next to it; fp32 MFMAs, v_mfma_f32_16x16x16_f16 or VALU work next to it are harmless.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_victim.hip -o /tmp/mfma_victim && /tmp/mfma_victim [aggressor workgroups] [victim workgroups] [launches]
//
// Aggressors: 0 dependent chains of v_mfma_f32_16x16x32_f16   1 the same on 6 independent accumulators   2 v_mfma_f32_16x16x16_f16 chains   3 v_mfma_f32_16x16x4_f32 chains
//             4 v_mfma_f32_16x16x32_bf16 chains   5 VALU only
// Victims:    0 back-to-back dependent chains of v_mfma_f32_16x16x4_f32 in registers   1 four independent fp32 MFMA accumulators   2 fp32 chains whose accumulator travels
//             through the LDS   3 VALU only (fma chains)   4 LDS only (write / read / add)   5 fp32 MFMAs whose B operand is read from the LDS right before each MFMA
//             6 DPP row_newbcast   7 LDS-DMA (global_load_lds)   8 packed fp32 VALU + v_readlane   9 LDS atomics + 128-bit LDS accesses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int mode>
__global__ __launch_bounds__(256) void aggressor_kernel(int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[6];
    for (int k = 0; k < 6; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 h0, hb;
    bf16x8 q0, qb;
    f16x4 k0, kb;
    for (int s = 0; s < 8; ++s) {
        h0[s] = (_Float16)(0.03f * (float)((lane + 3 * s) % 11) - 0.15f);
        hb[s] = (_Float16)(0.05f * (float)((lane * 3 + s) % 7) - 0.15f);
        q0[s] = (__bf16)(0.03f * (float)((lane + 3 * s) % 11) - 0.15f);
        qb[s] = (__bf16)(0.05f * (float)((lane * 3 + s) % 7) - 0.15f);
    }
    for (int s = 0; s < 4; ++s) { k0[s] = h0[s]; kb[s] = hb[s]; }
    const float a = 0.01f * (float)((lane * 7 + 3) % 17) - 0.08f, b = 0.02f * (float)((lane * 5 + 1) % 13) - 0.12f;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n s_nop 15\n s_nop 7" : "+v"(acc[0]) : "v"(h0), "v"(hb));
            acc[0] = acc[0] * 0.5f;
        } else if (mode == 1) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n v_mfma_f32_16x16x32_f16 %1, %6, %7, %1\n v_mfma_f32_16x16x32_f16 %2, %6, %7, %2\n"
                         "v_mfma_f32_16x16x32_f16 %3, %6, %7, %3\n v_mfma_f32_16x16x32_f16 %4, %6, %7, %4\n v_mfma_f32_16x16x32_f16 %5, %6, %7, %5\n s_nop 15\n s_nop 7"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]) : "v"(h0), "v"(hb));
            for (int k = 0; k < 6; ++k) acc[k] = acc[k] * 0.5f;
        } else if (mode == 2) {
            asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n"
                         "v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n s_nop 15\n s_nop 7" : "+v"(acc[0]) : "v"(k0), "v"(kb));
            acc[0] = acc[0] * 0.5f;
        } else if (mode == 3) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n"
                         "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n s_nop 15\n s_nop 15\n s_nop 7" : "+v"(acc[0]) : "v"(a), "v"(b));
            acc[0] = acc[0] * 0.5f;
        } else if (mode == 4) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n"
                         "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n s_nop 15\n s_nop 7" : "+v"(acc[0]) : "v"(q0), "v"(qb));
            acc[0] = acc[0] * 0.5f;
        } else {
            for (int k = 0; k < 6; ++k) acc[k] = acc[k] * 0.999f + a;
        }
    }
    float s = 0.f;
    for (int k = 0; k < 6; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ void chain_f32(f32x4& acc, float a0, float a1, float a2, float a3, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %2, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %3, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n"
                 "v_mfma_f32_16x16x4_f32 %0, %2, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %0, %3, %5, %0\n"
                 "s_nop 15\n s_nop 15\n s_nop 7"
                 : "+v"(acc) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));
}

template <int V>
__global__ __launch_bounds__(256) void victim_kernel(int iters, float* __restrict__ out, const float* __restrict__ gtab) {
    __shared__ float lds[4 * 1024 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float a0 = 0.01f * (float)((lane * 7 + 3) % 17) - 0.08f, a1 = 0.013f * (float)((lane * 3 + 5) % 19) - 0.11f, a2 = 0.017f * (float)((lane * 11 + 1) % 13) - 0.1f,
                a3 = 0.007f * (float)((lane * 5 + 7) % 23) - 0.07f, b = 0.02f * (float)((lane * 5 + 1) % 13) - 0.12f;
    f32x4 acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* tile = lds + wave * 1024;
    for (int r = 0; r < 16; ++r) tile[r * 64 + lane] = 0.01f * (float)((r * 5 + lane) % 23) - 0.1f;
    if (threadIdx.x < 4) reinterpret_cast<int*>(lds + 4 * 1024)[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (V == 0) {
            chain_f32(acc[0], a0, a1, a2, a3, b);
            acc[0] = acc[0] * 0.5f;
        } else if (V == 1) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, acc[3], 0, 0, 0);
            }
            for (int k = 0; k < 4; ++k) acc[k] = acc[k] * 0.5f;
        } else if (V == 2) {
            f32x4 t;
            for (int r = 0; r < 4; ++r) t[r] = tile[r * 64 + lane] * 0.5f;
            chain_f32(t, a0, a1, a2, a3, b);
            for (int r = 0; r < 4; ++r) tile[r * 64 + lane] = t[r];
        } else if (V == 3) {
            for (int k = 0; k < 4; ++k)
                for (int r = 0; r < 4; ++r) acc[k][r] = acc[k][r] * 0.75f + (a0 + 0.1f * k) * (b + 0.05f * r);
        } else if (V == 4) {
            for (int r = 0; r < 8; ++r) {
                const float x = tile[r * 64 + lane], y = tile[(r + 8) * 64 + (lane ^ 1)];
                tile[r * 64 + lane] = x * 0.5f + y * 0.25f + a0;
            }
        } else if (V == 6) {
            // DPP row_newbcast (the edge kernel's scale step: lane q of every row of 16 broadcast to the row), inline asm as there
            for (int k = 0; k < 4; ++k)
                for (int r = 0; r < 4; ++r) {
                    float o_;
                    const float v_ = acc[k][r] * 0.5f + a0 + 0.01f * r, x_ = b + 0.1f * k;
                    if (((k + r) & 3) == 0) asm volatile("s_nop 1\n v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(o_) : "v"(v_), "v"(x_));
                    else if (((k + r) & 3) == 1) asm volatile("s_nop 1\n v_mul_f32_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "=v"(o_) : "v"(v_), "v"(x_));
                    else if (((k + r) & 3) == 2) asm volatile("s_nop 1\n v_mul_f32_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf" : "=v"(o_) : "v"(v_), "v"(x_));
                    else asm volatile("s_nop 1\n v_mul_f32_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(o_) : "v"(v_), "v"(x_));
                    acc[k][r] = o_ + a1;
                }
        } else if (V == 7) {
            // LDS-DMA (global_load_lds, 16 and 4 bytes per lane: the edge kernel's staging of plain rows) from a small global table, read back and summed
            const float* gsrc = gtab + ((i & 7) * 256 + threadIdx.x) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)(tile), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int r = 0; r < 4; ++r) acc[0][r] = acc[0][r] * 0.5f + tile[lane * 4 + r];
        } else if (V == 8) {
            // packed fp32 VALU (v_pk_mul_f32 / v_pk_fma_f32) and v_readlane / readfirstlane
            for (int k = 0; k < 4; ++k) {
                acc[k] = acc[k] * (f32x4){0.5f, 0.25f, 0.5f, 0.25f} + (f32x4){a0, a1, a2, a3};
                const float sl = __builtin_amdgcn_readlane(__float_as_int(acc[k][0]), (i + k) & 63) == 0 ? 0.f : 1e-3f;
                acc[k][1] += sl;
            }
        } else if (V == 9) {
            // LDS atomics (the claim counter) + 128-bit LDS reads / writes
            if (lane == 0) atomicAdd(reinterpret_cast<int*>(lds + 4 * 1024) + wave, 1);
            f32x4 t = *reinterpret_cast<f32x4*>(tile + lane * 4);
            t = t * 0.5f + (f32x4){a0, a1, a2, a3};
            *reinterpret_cast<f32x4*>(tile + ((lane + 1) & 63) * 4) = t;
        } else {
            // B operand from the LDS right in front of every MFMA (the edge kernel's GEMM1 pattern), four accumulators
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float b0 = tile[(4 * q + 0) * 64 + lane], b1 = tile[(4 * q + 1) * 64 + lane], b2 = tile[(4 * q + 2) * 64 + lane], b3 = tile[(4 * q + 3) * 64 + lane];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc[3], 0, 0, 0);
            }
            for (int k = 0; k < 4; ++k) acc[k] = acc[k] * 0.5f;
        }
    }
    if (V == 2 || V == 4)
        for (int r = 0; r < 4; ++r) acc[0][r] = tile[r * 64 + lane];
    if (V == 9) {
        for (int r = 0; r < 4; ++r) acc[0][r] = tile[lane * 4 + r];
        acc[1][0] = (float)reinterpret_cast<int*>(lds + 4 * 1024)[wave];
    }
    float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 4; ++r) o[k * 4 + r] = acc[k][r];
}

static void launch_aggressor(int mode, int grid, int iters, float* out, hipStream_t s) {
    switch (mode) {
#define AG(M) case M: hipLaunchKernelGGL(aggressor_kernel<M>, dim3(grid), dim3(256), 0, s, iters, out); break;
        AG(0) AG(1) AG(2) AG(3) AG(4) AG(5)
    }
}
static float* g_tab = nullptr;
static void launch_victim(int v, int grid, int iters, float* out, hipStream_t s) {
    switch (v) {
#define VI(M) case M: hipLaunchKernelGGL(victim_kernel<M>, dim3(grid), dim3(256), 0, s, iters, out, g_tab); break;
        VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) VI(6) VI(7) VI(8) VI(9)
    }
}

int main(int argc, char** argv) {
    const int ag_grid = argc > 1 ? atoi(argv[1]) : 256, v_grid = argc > 2 ? atoi(argv[2]) : 4096, reps = argc > 3 ? atoi(argv[3]) : 5;
    const int v_iters = 3000;
    const size_t n = (size_t)v_grid * 256 * 16;
    float *d, *ag_out;
    CHECK(hipMalloc(&d, n * sizeof(float)));
    CHECK(hipMalloc(&ag_out, (size_t)ag_grid * 256 * sizeof(float)));
    {
        std::vector<float> tab(8 * 256 * 4);
        for (size_t i = 0; i < tab.size(); ++i) tab[i] = 0.001f * (float)((i * 37 + 11) % 211) - 0.1f;
        CHECK(hipMalloc(&g_tab, tab.size() * sizeof(float)));
        CHECK(hipMemcpy(g_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    hipStream_t sa, sv;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sv));
    std::vector<float> ref(n), got(n);
    const char* an[6] = {"dependent chains of v_mfma_f32_16x16x32_f16", "independent v_mfma_f32_16x16x32_f16", "chains of v_mfma_f32_16x16x16_f16", "chains of v_mfma_f32_16x16x4_f32", "chains of v_mfma_f32_16x16x32_bf16", "VALU only"};
    const char* vn[10] = {"back-to-back dependent fp32 MFMA chains (registers)", "independent fp32 MFMAs (registers)", "fp32 MFMA chains through the LDS", "VALU only", "LDS only", "fp32 MFMAs with B operands from the LDS", "DPP row_newbcast", "LDS-DMA (global_load_lds)", "packed fp32 VALU + v_readlane", "LDS atomics + 128-bit LDS accesses"};
    for (int v = (argc > 4 ? atoi(argv[4]) : 0); v < 10; ++v) {
        launch_victim(v, v_grid, v_iters, d, sv);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(ref.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        // how long the victim runs alone -> aggressor iterations for about 3 x that
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, sv)); launch_victim(v, v_grid, v_iters, d, sv); CHECK(hipEventRecord(e1, sv)); CHECK(hipDeviceSynchronize());
        float v_ms = 0.f; CHECK(hipEventElapsedTime(&v_ms, e0, e1));
        for (int mode : {3, 0, 1, 4}) {
            CHECK(hipEventRecord(e0, sa)); launch_aggressor(mode, ag_grid, 20000, ag_out, sa); CHECK(hipEventRecord(e1, sa)); CHECK(hipDeviceSynchronize());
            float a_ms = 0.f; CHECK(hipEventElapsedTime(&a_ms, e0, e1));
            const int a_iters = (int)(20000.0 * (4.0 * v_ms + 20.0) / (a_ms > 1e-3f ? a_ms : 1e-3f)) + 1000;
            long bad_runs = 0, bad_wgs = 0;
            double worst = 0.0;
            for (int rep = 0; rep < reps; ++rep) {
                CHECK(hipMemset(d, 0, n * sizeof(float)));
                CHECK(hipDeviceSynchronize());
                launch_aggressor(mode, ag_grid, a_iters, ag_out, sa);
                launch_victim(v, v_grid, v_iters, d, sv);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(got.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
                long bw = 0;
                for (int wg = 0; wg < v_grid; ++wg) {
                    const size_t o = (size_t)wg * 256 * 16;
                    if (memcmp(&got[o], &ref[o], 256 * 16 * sizeof(float))) {
                        ++bw;
                        for (size_t i = o; i < o + 256 * 16; ++i) { double e = fabs((double)got[i] - ref[i]) / (fabs((double)ref[i]) + 1e-30); if (e > worst && ref[i] != 0.f) worst = e; }
                    }
                }
                bad_wgs += bw;
                bad_runs += bw > 0;
            }
            printf("{\"victim\": \"%s\", \"aggressor\": \"%s\", \"aggressor_workgroups\": %d, \"launches\": %d, \"launches_with_wrong_workgroups\": %ld, \"wrong_victim_workgroups\": %ld, \"of\": %ld, \"worst_rel\": %.3e, \"victim_ms_alone\": %.2f}\n",
                   vn[v], an[mode], ag_grid, reps, bad_runs, bad_wgs, (long)v_grid * reps, worst, v_ms);
            fflush(stdout);
        }
    }
    return 0;
}

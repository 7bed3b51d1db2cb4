// mfma_war.hip -- standalone probe (no part of the library): WRITE-AFTER-READ on MFMA operand registers, gfx950.
// Background: profiles/r06_tp_is.md sections 3 / 4.  tools/mfma_mix.hip ran the two MFMA kinds side by side on CONSTANT operands and was clean; what it never did is
// what the edge kernel does all the time: hand an MFMA's A operand register to a load (or a VALU write) directly behind the MFMA.  If an issued MFMA can wait in the
// matrix pipe's queue (a partner wave's MFMAs ahead of it) and reads its operands only when it starts, a fast load return can overwrite them first.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_war.hip -o /tmp/mfma_war && /tmp/mfma_war [workgroups] [iterations] [launches]
//
// Every wave runs `iters` rounds of: [optional half-precision MFMAs] -> one MFMA that reads operand register R -> GAP -> a write to R (LDS read / global load / VALU)
// with a NEW value (two value sets, toggled per round) -> wait for it -> next round.  GAP = nothing ("tight") or 8 x s_nop 15 ("safe").  The reference result is the
// safe form with one wave per SIMD (160 KB of LDS per workgroup); every other (form, waves per SIMD) must reproduce it bit for bit.
//   mode 0  fp32 MFMA reads A, LDS read into A behind it                      (control: the shipped kernel's pattern)
//   mode 1  3 half-precision MFMAs, then the fp32 MFMA, LDS read into ITS A   (queued behind the half-precision ones?)
//   mode 2  as 1, global load instead of the LDS read
//   mode 3  half-precision MFMA, global load into ITS A (4 registers)          (section 3's finding, in isolation)
//   mode 4  half-precision MFMA, LDS read (b128) into ITS A
//   mode 5  as 1, VALU write instead of the LDS read
// Odd workgroups of modes >= 10 (mode - 10 = the above) run plain fp32 MFMA bursts instead: the partner wave keeps the pipe as busy as it can.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define NOP8 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
#define H3 "v_mfma_f32_16x16x32_f16 %[h0], %[a8], %[b8], %[h0]\n v_mfma_f32_16x16x32_f16 %[h1], %[a8], %[b8], %[h1]\n v_mfma_f32_16x16x32_f16 %[h2], %[a8], %[b8], %[h2]\n"
#define F1 "v_mfma_f32_16x16x4_f32 %[f], %[va], %[vb], %[f]\n"
#define TAIL "s_nop 15\n s_nop 15\n s_waitcnt vmcnt(0) lgkmcnt(0)\n"

struct St {
    f32x4 f, h0, h1, h2;
    f16x8 a8, b8;
    float va, vb;
};

// one round; la = LDS byte address of this lane's new scalar / vector value, ga = global address of the same
template <int MODE, bool SAFE>
__device__ __forceinline__ void round_(St& s, unsigned la, const float* ga, float vnew) {
    if (MODE == 0) {
        if (SAFE) asm volatile(F1 NOP8 "ds_read_b32 %[va], %[la]\n" TAIL : [f] "+v"(s.f), [va] "+v"(s.va) : [vb] "v"(s.vb), [la] "v"(la) : "memory");
        else asm volatile(F1 "ds_read_b32 %[va], %[la]\n" TAIL : [f] "+v"(s.f), [va] "+v"(s.va) : [vb] "v"(s.vb), [la] "v"(la) : "memory");
    } else if (MODE == 1) {
        if (SAFE) asm volatile(H3 F1 NOP8 "ds_read_b32 %[va], %[la]\n" TAIL : [f] "+v"(s.f), [va] "+v"(s.va), [h0] "+v"(s.h0), [h1] "+v"(s.h1), [h2] "+v"(s.h2)
                               : [vb] "v"(s.vb), [la] "v"(la), [a8] "v"(s.a8), [b8] "v"(s.b8) : "memory");
        else asm volatile(H3 F1 "ds_read_b32 %[va], %[la]\n" TAIL : [f] "+v"(s.f), [va] "+v"(s.va), [h0] "+v"(s.h0), [h1] "+v"(s.h1), [h2] "+v"(s.h2)
                          : [vb] "v"(s.vb), [la] "v"(la), [a8] "v"(s.a8), [b8] "v"(s.b8) : "memory");
    } else if (MODE == 2) {
        if (SAFE) asm volatile(H3 F1 NOP8 "global_load_dword %[va], %[ga], off\n" TAIL : [f] "+v"(s.f), [va] "+v"(s.va), [h0] "+v"(s.h0), [h1] "+v"(s.h1), [h2] "+v"(s.h2)
                               : [vb] "v"(s.vb), [ga] "v"(ga), [a8] "v"(s.a8), [b8] "v"(s.b8) : "memory");
        else asm volatile(H3 F1 "global_load_dword %[va], %[ga], off\n" TAIL : [f] "+v"(s.f), [va] "+v"(s.va), [h0] "+v"(s.h0), [h1] "+v"(s.h1), [h2] "+v"(s.h2)
                          : [vb] "v"(s.vb), [ga] "v"(ga), [a8] "v"(s.a8), [b8] "v"(s.b8) : "memory");
    } else if (MODE == 3) {
        if (SAFE) asm volatile("v_mfma_f32_16x16x32_f16 %[h0], %[a8], %[b8], %[h0]\n" NOP8 "global_load_dwordx4 %[a8], %[ga], off\n" TAIL : [h0] "+v"(s.h0), [a8] "+v"(s.a8)
                               : [ga] "v"(ga), [b8] "v"(s.b8) : "memory");
        else asm volatile("v_mfma_f32_16x16x32_f16 %[h0], %[a8], %[b8], %[h0]\n global_load_dwordx4 %[a8], %[ga], off\n" TAIL : [h0] "+v"(s.h0), [a8] "+v"(s.a8)
                          : [ga] "v"(ga), [b8] "v"(s.b8) : "memory");
    } else if (MODE == 4) {
        if (SAFE) asm volatile("v_mfma_f32_16x16x32_f16 %[h0], %[a8], %[b8], %[h0]\n" NOP8 "ds_read_b128 %[a8], %[la]\n" TAIL : [h0] "+v"(s.h0), [a8] "+v"(s.a8)
                               : [la] "v"(la), [b8] "v"(s.b8) : "memory");
        else asm volatile("v_mfma_f32_16x16x32_f16 %[h0], %[a8], %[b8], %[h0]\n ds_read_b128 %[a8], %[la]\n" TAIL : [h0] "+v"(s.h0), [a8] "+v"(s.a8)
                          : [la] "v"(la), [b8] "v"(s.b8) : "memory");
    } else {
        if (SAFE) asm volatile(H3 F1 NOP8 "v_mov_b32 %[va], %[nv]\n" TAIL : [f] "+v"(s.f), [va] "+v"(s.va), [h0] "+v"(s.h0), [h1] "+v"(s.h1), [h2] "+v"(s.h2)
                               : [vb] "v"(s.vb), [nv] "v"(vnew), [a8] "v"(s.a8), [b8] "v"(s.b8) : "memory");
        else asm volatile(H3 F1 "v_mov_b32 %[va], %[nv]\n" TAIL : [f] "+v"(s.f), [va] "+v"(s.va), [h0] "+v"(s.h0), [h1] "+v"(s.h1), [h2] "+v"(s.h2)
                          : [vb] "v"(s.vb), [nv] "v"(vnew), [a8] "v"(s.a8), [b8] "v"(s.b8) : "memory");
    }
}

template <int MODE, bool SAFE>
__device__ __forceinline__ void run_(St& s, int iters, unsigned lds_lane, const float* glane, int lane) {
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        const int set = i & 1;
        // the two value sets sit 4 KB (LDS) / 4 KB (global) apart; the accumulators are damped so that they stay finite
        round_<MODE, SAFE>(s, lds_lane + set * 4096, glane + set * 1024, set ? 0.0625f + 0.001f * lane : -0.03125f + 0.002f * lane);
        asm volatile("s_nop 7\n v_mul_f32 %0, 0.75, %0\n v_mul_f32 %1, 0.75, %1\n v_mul_f32 %2, 0.75, %2\n v_mul_f32 %3, 0.75, %3" : "+v"(s.f[0]), "+v"(s.f[1]), "+v"(s.f[2]), "+v"(s.f[3]));
        asm volatile("v_mul_f32 %0, 0.75, %0\n v_mul_f32 %1, 0.75, %1\n v_mul_f32 %2, 0.75, %2\n v_mul_f32 %3, 0.75, %3" : "+v"(s.h0[0]), "+v"(s.h0[1]), "+v"(s.h0[2]), "+v"(s.h0[3]));
    }
}

__device__ __forceinline__ void burst_f32(f32x4 (&acc)[4], float a, float b, int n) {
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + 0.125f * k, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = acc[k] * 0.75f;
    }
}

__global__ __launch_bounds__(256, 2) void war_kernel(int mode_in, int safe, int iters, const float* __restrict__ gvals, float* __restrict__ out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const bool partner = mode_in >= 10 && (blockIdx.x & 1);
    const int mode = mode_in >= 10 ? mode_in - 10 : mode_in;
    // LDS: two value sets of 256 lanes x 4 floats
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = gvals[i];
    __syncthreads();
    St s;
    s.f = s.h0 = s.h1 = s.h2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 8; ++k) { s.a8[k] = (_Float16)(0.03f * (float)((lane + 3 * k) % 11) - 0.15f); s.b8[k] = (_Float16)(0.05f * (float)((lane * 3 + k) % 7) - 0.15f); }
    s.va = 0.01f * (float)((lane * 7 + 3) % 17) - 0.08f;
    s.vb = 0.02f * (float)((lane * 5 + 1) % 13) - 0.12f;
    float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (partner) {
        f32x4 acc[4];
        for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        burst_f32(acc, s.va, s.vb, iters);
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < 4; ++r) o[k * 4 + r] = acc[k][r];
        return;
    }
    const unsigned lds_lane = (unsigned)(threadIdx.x * 16);       // byte address of this lane's float4 in set 0 (the dynamic LDS starts at 0)
    const float* glane = gvals + threadIdx.x * 4;
#define RUN(M) case M: if (safe) run_<M, true>(s, iters, lds_lane, glane, lane); else run_<M, false>(s, iters, lds_lane, glane, lane); break;
    switch (mode) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    for (int r = 0; r < 4; ++r) { o[r] = s.f[r]; o[4 + r] = s.h0[r]; o[8 + r] = s.h1[r]; o[12 + r] = s.h2[r]; }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 2000, reps = argc > 3 ? atoi(argv[3]) : 10;
    const size_t n = (size_t)grid * 256 * 16;
    float *d, *gv;
    CHECK(hipMalloc(&d, n * sizeof(float)));
    CHECK(hipMalloc(&gv, 2048 * sizeof(float)));
    std::vector<float> vals(2048);
    for (int i = 0; i < 2048; ++i) vals[i] = 0.001f * (float)((i * 37 + 11) % 211) - 0.1f;
    // (as half-precision pairs the same bits are finite small numbers or harmless: keep them away from inf / nan patterns)
    for (int i = 0; i < 2048; ++i) { unsigned u; memcpy(&u, &vals[i], 4); u &= 0xBBFFBBFFu; memcpy(&vals[i], &u, 4); }
    CHECK(hipMemcpy(gv, vals.data(), 2048 * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void*)war_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<float> ref(n), got(n);
    const char* names[6] = {"fp32 MFMA, LDS read into its A", "3 f16 MFMAs + fp32 MFMA, LDS read into the fp32 MFMA's A", "3 f16 MFMAs + fp32 MFMA, global load into the fp32 MFMA's A",
                            "f16 MFMA, global load (x4) into its A", "f16 MFMA, LDS read (b128) into its A", "3 f16 MFMAs + fp32 MFMA, VALU write into the fp32 MFMA's A"};
    for (int base : {0, 10})
        for (int m = 0; m < 6; ++m) {
            const int mode = base + m;
            hipLaunchKernelGGL(war_kernel, dim3(grid), dim3(256), 160 * 1024, 0, mode, 1, iters, gv, d);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(ref.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
            for (int safe : {1, 0})
                for (int lds_kb : {160, 80}) {
                    long bad_runs = 0, bad_wgs = 0;
                    double worst = 0.0;
                    for (int rep = 0; rep < reps; ++rep) {
                        CHECK(hipMemset(d, 0, n * sizeof(float)));
                        hipLaunchKernelGGL(war_kernel, dim3(grid), dim3(256), lds_kb * 1024, 0, mode, safe, iters, gv, d);
                        CHECK(hipDeviceSynchronize());
                        CHECK(hipMemcpy(got.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
                        long bw = 0;
                        for (int wg = 0; wg < grid; ++wg) {
                            const size_t o = (size_t)wg * 256 * 16;
                            if (memcmp(&got[o], &ref[o], 256 * 16 * sizeof(float))) {
                                ++bw;
                                for (size_t i = o; i < o + 256 * 16; ++i) { double e = fabs((double)got[i] - ref[i]) / (fabs((double)ref[i]) + 1e-30); if (e > worst && ref[i] != 0.f) worst = e; }
                            }
                        }
                        bad_wgs += bw;
                        bad_runs += bw > 0;
                    }
                    printf("{\"mode\": %d, \"what\": \"%s%s\", \"gap\": \"%s\", \"waves_per_simd\": %d, \"launches\": %d, \"launches_with_wrong_workgroups\": %ld, \"wrong_workgroups\": %ld, \"of\": %ld, \"worst_rel\": %.3e}\n",
                           mode, names[m], base ? " | odd workgroups: fp32 MFMA bursts" : "", safe ? "8 x s_nop 15" : "none", lds_kb == 160 ? 1 : 2, reps, bad_runs, bad_wgs, (long)grid * reps, worst);
                    fflush(stdout);
                }
        }
    CHECK(hipFree(d));
    CHECK(hipFree(gv));
    return 0;
}

// xdl_aggressor.hip -- a SEPARATE kernel that does nothing but issue MFMAs in registers (no LDS, no memory traffic, <= 64 VGPRs: one of its waves fits on a SIMD next to
// two waves of the edge kernel), launched on a side stream while the UNMODIFIED shipped library computes (tools/gpu_aggressor.py; profiles/r06_tp_is.md section 8).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tests/csrc/xdl_aggressor.hip -o /tmp/libxdl_aggressor.so
// mode 0: dependent chains of v_mfma_f32_16x16x32_f16   1: the same MFMAs on 6 independent accumulators   2: dependent chains of v_mfma_f32_16x16x16_f16
// mode 3: dependent chains of v_mfma_f32_16x16x4_f32 (control)   4: dependent chains of v_mfma_f32_16x16x32_bf16   5: no MFMA at all (VALU only, control)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

static float* g_out = nullptr;
extern "C" int aggressor_launch(int mode, int grid, int iters, void* stream) {
    if (!g_out && hipMalloc(&g_out, (size_t)65536 * 256 * sizeof(float)) != hipSuccess) return -1;
    if (grid > 65536) return -2;
    switch (mode) {
#define AG(M) case M: hipLaunchKernelGGL(aggressor_kernel<M>, dim3(grid), dim3(256), 0, (hipStream_t)stream, iters, g_out); break;
        AG(0) AG(1) AG(2) AG(3) AG(4) AG(5)
        default: return -4;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

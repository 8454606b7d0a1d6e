// MFMA issue cost on gfx950 (wave64), 4 waves per SIMD, 8 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 16384
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int OP> __global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    f32x4 acc[8]; f32x16 big[2];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) big[i][j] = seed;
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
    f16x8 ha, hb; for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)a; hb[j] = (_Float16)b; }
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0); }
        if (OP == 1) { for (int i = 0; i < 2; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i], 0, 0, 0); }
        if (OP == 2) { for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0); }
        if (OP == 3) {  // 8 MFMA f32 + 16 independent FMAs: do they overlap?
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2" : "+v"(v[i]), "+v"(v[(i + 4) & 7]) : "v"(a), "v"(b), "v"(v[(i+1)&7]));
            }
        }
        if (OP == 4) {  // same FMAs without the MFMAs
            for (int i = 0; i < 8; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2" : "+v"(v[i]), "+v"(v[(i + 4) & 7]) : "v"(a), "v"(b), "v"(v[(i+1)&7]));
        }
        if (OP == 5) {  // f16 MFMA + FMAs
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2" : "+v"(v[i]), "+v"(v[(i + 4) & 7]) : "v"(a), "v"(b), "v"(v[(i+1)&7]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i]; s += big[0][0] + big[1][5];
    if (s == 12345.678f) out[0] = s;
}
template <int OP> void run(const char* name, float* out, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, 1024>>>(out, ITERS, 1.0f);
    hipEventRecord(e0);
    k<OP><<<256, 1024>>>(out, ITERS, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.3f ms  %7.2f cycles per instruction per wave (4 waves/SIMD, 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (4.0 * ITERS * n));
}
int main() {
    float* out; hipMalloc(&out, 64);
    run<0>("warm", out, 8); run<0>("warm", out, 8);
    run<0>("v_mfma_f32_16x16x4_f32", out, 8); run<1>("v_mfma_f32_32x32x2_f32", out, 2); run<2>("v_mfma_f32_16x16x32_f16", out, 8);
    run<4>("16 v_fma_f32", out, 16); run<3>("8 mfma f32 + 16 fma (per mfma)", out, 8); run<5>("8 mfma f16 + 16 fma (per mfma)", out, 8);
    return 0;
}

// Issue rate of the f16 MFMA shapes a weight-gradient product could be built from, one wave per SIMD, four
// independent accumulators, every CU: v_mfma_f32_16x16x32_f16 (gfx950's K = 32 form), the older
// v_mfma_f32_16x16x16_f16 (K = 16: operands of two registers, no repeated halves needed) and v_mfma_f32_16x16x4_f32.
//   hipcc --offload-arch=gfx950 -O3 -o ubench5 ubench5.hip && ./ubench5
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 65536
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, 0, 0, 0};
    f16x8 a8, b8; f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(seed + i + threadIdx.x * 1e-3f); b8[i] = (_Float16)(1e-3f * i); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    const float af = seed, bf = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (OP == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
                if (OP == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
                if (OP == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
            }
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
template <int OP> void run(const char* name, float* out, double mhz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k<OP><<<256, 256>>>(out, ITERS, 1.0f);   // ramp clocks
    hipEventRecord(e0);
    k<OP><<<256, 256>>>(out, ITERS, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %6.2f cycles per MFMA per SIMD (at %.0f MHz nominal)\n", name, ms, ms * 1e-3 * mhz * 1e6 / (ITERS * 16.0), mhz);
}
int main() {
    float* out; hipMalloc(&out, 64);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double mhz = khz / 1000.0;
    run<0>("v_mfma_f32_16x16x32_f16", out, mhz); run<0>("v_mfma_f32_16x16x32_f16", out, mhz);
    run<1>("v_mfma_f32_16x16x16_f16", out, mhz); run<2>("v_mfma_f32_16x16x4_f32", out, mhz);
    run<0>("v_mfma_f32_16x16x32_f16", out, mhz);
    return 0;
}

// icache.hip -- does straight-line code beyond the 64 KB instruction cache slow a 1-wave-per-SIMD MFMA loop?
// Body: REP x { v_mfma_f32_16x16x32_f16 (8 B) + 2 v_fma_f32 (8 B each) } = 24 REP bytes, looped `iters` times.
// Build: hipcc --offload-arch=gfx950 -O3 -o icache icache.hip ; run: ./icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int REP>
__global__ __launch_bounds__(256) void body(float* out, int iters) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float x = threadIdx.x, y = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_fma_f32 %1, %1, %4, %1\n\tv_fma_f32 %4, %4, %1, %4"
                         : "+v"(c0), "+v"(x), "+v"(a), "+v"(b), "+v"(y));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + x + y;
}

template <int REP>
void run(float* d, int total_rep) {
    const int iters = total_rep / REP;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    body<REP><<<256, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    body<REP><<<256, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("body %6.1f KB  iters %6d  %.3f ms  %.2f ns per MFMA group\n", REP * 24 / 1024.0, iters, ms,
           ms * 1e6 / ((double)iters * REP));
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    const int total = 1 << 21;
    run<256>(d, total); run<1024>(d, total); run<2048>(d, total); run<2560>(d, total); run<2816>(d, total);
    run<3072>(d, total); run<4096>(d, total); run<6144>(d, total); run<8192>(d, total);
    return 0;
}

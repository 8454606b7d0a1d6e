// HBM write rate of the two store patterns the [N][64] float32 row tensors are written with (1 GiB per launch):
//   rows64: a wave's 16-byte stores cover 16 rows x 64 contiguous bytes per instruction (lane (i, g): row i, bytes
//           64 m + 16 g of the row; four instructions m = 0..3 complete the 16 rows) -- the MFMA operand layout
//   full  : a wave's 16-byte stores cover 1,024 contiguous bytes per instruction (four whole rows)
// and the read rate of the same two patterns.
//   hipcc --offload-arch=gfx950 -O3 -o ubench6 ubench6.hip && ./ubench6
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT, bool READ> __global__ __launch_bounds__(256) void k(float4* buf, long ntile, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (long t = (long)blockIdx.x * 4 + wave; t < ntile; t += (long)gridDim.x * 4) {
        float4* base = buf + t * 256;   // 16 rows of 16 float4
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float4* p = PAT == 0 ? base + i * 16 + 4 * m + g : base + m * 64 + lane;
            if (READ) { float4 v = *p; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            else *p = make_float4((float)t, (float)m, (float)lane, 1.0f);
        }
    }
    if (READ && acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
template <int PAT, bool READ> void run(const char* name, float4* buf, long ntile, float* sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k<PAT, READ><<<2048, 256>>>(buf, ntile, sink);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<PAT, READ><<<2048, 256>>>(buf, ntile, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %7.3f ms per GiB  %6.2f TB/s\n", name, ms / 5, 5.0 * ntile * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    const long ntile = 262144;   // x 4 KiB = 1 GiB
    float4* buf; float* sink; (void)hipMalloc(&buf, ntile * 4096); (void)hipMalloc(&sink, 64);
    run<0, false>("write rows64", buf, ntile, sink); run<1, false>("write full", buf, ntile, sink);
    run<0, false>("write rows64", buf, ntile, sink); run<1, false>("write full", buf, ntile, sink);
    run<0, true>("read rows64", buf, ntile, sink); run<1, true>("read full", buf, ntile, sink);
    return 0;
}

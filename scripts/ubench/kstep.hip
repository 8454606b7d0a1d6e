// kstep.hip -- what one k-step of wide_fused_kernel costs on one wave per SIMD: three dependent
// v_mfma_f32_32x32x16_f16 on one accumulator, two ds_read_b128 two steps ahead, a few VALU instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -o kstep kstep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    f16x8 b0, b1;
    for (int j = 0; j < 8; ++j) { b0[j] = (_Float16)(0.001f * threadIdx.x + j); b1[j] = (_Float16)(j * 0.5f); }
    f32x16 acc = {}, acc2 = {};
    float x = threadIdx.x, y = 1.0f, z = 2.0f;
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + 16 * (threadIdx.x & 63);
    u32x4 wa, wb, wc, wd;
    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                 : "=v"(wa), "=v"(wb), "=v"(wc), "=v"(wd) : "v"(addr));
    for (int it = 0; it < iters; ++it) {
        u32x4 na = wa, nb = wb;
        if (MODE & 1) asm volatile("ds_read_b128 %0, %2 offset:4096\n ds_read_b128 %1, %2 offset:5120" : "=v"(na), "=v"(nb) : "v"(addr));
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 h = __builtin_bit_cast(f16x8, wa), l = __builtin_bit_cast(f16x8, wb);
        if (MODE & 4) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, b0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, b1, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(l, b0, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(l, b0, acc, 0, 0, 0);
        }
        if (MODE & 2) {
            x = fmaf(x, y, z); y = fmaxf(y, x); z = fmaf(z, 0.5f, x); x = fmaf(x, 0.25f, y); y = fmaf(y, z, x); z = fmaxf(z, y);
        }
        if (MODE & 8) {
            x = fmaf(x, y, z); y = fmaxf(y, x); z = fmaf(z, 0.5f, x); x = fmaf(x, 0.25f, y); y = fmaf(y, z, x); z = fmaxf(z, y);
            x = fmaf(x, y, z); y = fmaxf(y, x); z = fmaf(z, 0.5f, x); x = fmaf(x, 0.25f, y); y = fmaf(y, z, x); z = fmaxf(z, y);
        }
        for (int q = 0; q < 3; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wc), "+v"(wd));
        wa = wc; wb = wd; wc = na; wd = nb;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc2[1] + x + y + z + __builtin_bit_cast(float, wa[0]);
}
template <int MODE>
void run(float* d, const char* what) {
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<256, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %.1f ns per k-step (96 MFMA cycles = 40.0 ns at 2.4 GHz)\n", what, ms * 1e6 / iters);
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<0>(d, "3 dependent MFMA 32x32x16");
    run<4>(d, "3 MFMA on two accumulators");
    run<1>(d, "3 dependent MFMA + 2 ds_read_b128 + lgkmcnt(2)");
    run<3>(d, "... + 6 dependent VALU");
    run<11>(d, "... + 18 dependent VALU");
    run<2>(d, "3 dependent MFMA + 6 dependent VALU");
    return 0;
}

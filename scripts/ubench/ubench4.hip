// Issue cost of the packed / 16-bit vector instructions a reduced-precision sampling loop would be built from
// (BASELINE config 5's "bf16 forward-model math": gfx950 has no bf16 vector arithmetic, f16 is what there is), beside
// v_fma_f32; same harness as ubench.hip: 4 waves per SIMD, 16 independent instructions per trip, every CU.
//   hipcc --offload-arch=gfx950 -O3 -o ubench4 ubench4.hip && ./ubench4
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 131072
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP> __global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    float a[8]; unsigned u[8], h[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; u[i] = (unsigned)(threadIdx.x * 2654435761u + i); h[i] = 0x3c003800u + i; }
    const float c1 = 0.999f, c2 = 1e-3f; const unsigned hc = 0x3bff3bffu, hd = 0x14001400u;
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define PKFMAH(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(h[i]) : "v"(hc), "v"(hd));
#define PKADDH(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(h[i]) : "v"(hd));
#define FMAH(i) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(h[i]) : "v"(hc), "v"(hd));
#define EXPH(i) asm volatile("v_exp_f16 %0, %0" : "+v"(h[i]));
#define MIX(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h[i]), "v"(c2));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(u[i]) : "s"(127u));
#define CVTSD(i) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(a[i]) : "v"(u[i]));
#define DOT2(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(h[i]), "v"(hc));
        if (OP == 0) { REP8(FMA) REP8(FMA) }
        if (OP == 1) { REP8(PKFMAH) REP8(PKFMAH) }
        if (OP == 2) { REP8(PKADDH) REP8(PKADDH) }
        if (OP == 3) { REP8(FMAH) REP8(FMAH) }
        if (OP == 4) { REP8(EXPH) REP8(EXPH) }
        if (OP == 5) { REP8(MIX) REP8(MIX) }
        if (OP == 6) { REP8(ALIGN) REP8(ALIGN) }
        if (OP == 7) { REP8(CVTSD) REP8(CVTSD) }
        if (OP == 8) { REP8(DOT2) REP8(DOT2) }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i] + (float)h[i];
    if (s == 12345.678f) out[0] = s;
}
template <int OP> void run(const char* name, float* out, double mhz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k<OP><<<256, 1024>>>(out, ITERS, 1.0f);   // ramp clocks
    hipEventRecord(e0);
    k<OP><<<256, 1024>>>(out, ITERS, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-26s %8.3f ms  %6.2f cycles/instr/wave (at %.0f MHz)\n", name, ms, ms * 1e-3 * mhz * 1e6 / (4.0 * ITERS * 16), mhz);
}
int main() {
    float* out; hipMalloc(&out, 64);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double mhz = khz / 1000.0;
    run<0>("v_fma_f32", out, mhz); run<0>("v_fma_f32", out, mhz); run<1>("v_pk_fma_f16", out, mhz); run<2>("v_pk_add_f16", out, mhz);
    run<3>("v_fma_f16", out, mhz); run<4>("v_exp_f16", out, mhz); run<5>("v_fma_mix_f32", out, mhz);
    run<6>("v_alignbit_b32 (sgpr)", out, mhz); run<7>("v_cvt_f32_u32_sdwa", out, mhz); run<8>("v_dot2c_f32_f16", out, mhz);
    run<0>("v_fma_f32", out, mhz);
    return 0;
}

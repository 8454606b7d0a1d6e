// per-instruction issue cost on gfx950 (wave64), 4 waves per SIMD, independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 131072
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP> __global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    float a[8]; unsigned u[8]; unsigned long long w[8]; double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; u[i] = (unsigned)(threadIdx.x * 2654435761u + i); w[i] = u[i]; d[i] = a[i]; }
    float2 p[8]; for (int i = 0; i < 8; ++i) p[i] = make_float2(a[i], a[i] + 1);
    const float c1 = 0.999f, c2 = 1e-3f; const unsigned m = 0xD2511F53u;
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i+1)&7]), "v"(p[(i+2)&7]));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(u[i]), "s"(m) : "vcc"); 
#define MAD64S(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, 0" : "=v"(w[i]) : "v"(u[i]), "s"(m) : "s20","s21"); 
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define SIN(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define CVTI(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
#define CVTF(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define BITOP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "v"(u[(i+1)&7]), "s"(m));
#define FRACT(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "s"(m));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "s"(m));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i+1)&7]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i+1)&7]));
#define FMAMK(i) asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a[i]) : "v"(c2));
#define COS(i) asm volatile("v_cos_f32 %0, %0" : "+v"(a[i]));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define MULF(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
#define FMA3(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i+1)&7]), "v"(a[(i+2)&7]), "v"(a[(i+3)&7]));
#define FMAS(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(c2));
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
#define MIXET(i) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3" : "+v"(a[i&3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]) : "v"(c1), "v"(c2) : );
        if (OP == 0) { REP8(FMA) REP8(FMA) }
        if (OP == 1) { REP8(PKFMA) REP8(PKFMA) }
        if (OP == 2) { REP8(MAD64) REP8(MAD64) }
        if (OP == 3) { REP8(EXP) REP8(EXP) }
        if (OP == 4) { REP8(LOG) REP8(LOG) }
        if (OP == 5) { REP8(RCP) REP8(RCP) }
        if (OP == 6) { REP8(SIN) REP8(SIN) }
        if (OP == 7) { REP8(SQRT) REP8(SQRT) }
        if (OP == 8) { REP8(CVTI) REP8(CVTI) }
        if (OP == 9) { REP8(CVTF) REP8(CVTF) }
        if (OP == 10) { REP8(BITOP) REP8(BITOP) }
        if (OP == 11) { REP8(FRACT) REP8(FRACT) }
        if (OP == 12) { REP8(MULLO) REP8(MULLO) }
        if (OP == 13) { REP8(MULHI) REP8(MULHI) }
        if (OP == 14) { REP8(MUL24) REP8(MUL24) }
        if (OP == 15) { REP8(MED3) REP8(MED3) }
        if (OP == 16) { REP8(LSHLADD) REP8(LSHLADD) }
        if (OP == 17) { REP8(FMA64) REP8(FMA64) }
        if (OP == 18) { REP8(PKMUL) REP8(PKMUL) }
        if (OP == 19) { REP8(FMAMK) REP8(FMAMK) }
        if (OP == 20) { REP8(XOR) REP8(XOR) }
        if (OP == 21) { MIXET(0) MIXET(1) MIXET(2) MIXET(3) }   // 4 exp + 12 fma: do trans ops hide behind fma?
        if (OP == 23) { REP8(FMAC) REP8(FMAC) }
        if (OP == 24) { REP8(ADDF) REP8(ADDF) }
        if (OP == 25) { REP8(MULF) REP8(MULF) }
        if (OP == 26) { REP8(FMA3) REP8(FMA3) }
        if (OP == 27) { REP8(FMAS) REP8(FMAS) }
        if (OP == 22) { REP8(MAD64S) REP8(MAD64S) }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i] + (float)w[i] + (float)d[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int OP> void run(const char* name, float* out, double mhz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256;
    k<OP><<<blocks, 1024>>>(out, 64, 1.0f);
    for (int r = 0; r < 3; ++r) k<OP><<<blocks, 1024>>>(out, ITERS, 1.0f);   // ramp clocks
    hipEventRecord(e0);
    k<OP><<<blocks, 1024>>>(out, ITERS, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x ITERS x 16 instructions
    const double cyc = ms * 1e-3 * mhz * 1e6 / (4.0 * ITERS * 16);
    printf("%-22s %8.3f ms  %6.2f cycles/instr/wave (at %.0f MHz)\n", name, ms, cyc, mhz);
}
int main() {
    float* out; hipMalloc(&out, 64);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double mhz = khz / 1000.0;
    run<0>("v_fma_f32", out, mhz); run<0>("v_fma_f32", out, mhz); run<1>("v_pk_fma_f32", out, mhz);
    run<2>("v_mad_u64_u32 vcc", out, mhz); run<22>("v_mad_u64_u32 sgpr", out, mhz);
    run<3>("v_exp_f32", out, mhz); run<4>("v_log_f32", out, mhz);
    run<5>("v_rcp_f32", out, mhz); run<6>("v_sin_f32", out, mhz); run<7>("v_sqrt_f32", out, mhz); run<8>("v_cvt_i32_f32", out, mhz);
    run<9>("v_cvt_f32_u32", out, mhz); run<10>("v_bitop3_b32", out, mhz); run<11>("v_fract_f32", out, mhz); run<12>("v_mul_lo_u32", out, mhz);
    run<13>("v_mul_hi_u32", out, mhz); run<14>("v_mul_u32_u24", out, mhz); run<15>("v_med3_f32", out, mhz); run<16>("v_lshl_add_u32", out, mhz);
    run<17>("v_fma_f64", out, mhz); run<18>("v_pk_mul_f32", out, mhz); run<19>("v_fmamk_f32", out, mhz); run<20>("v_xor_b32", out, mhz);
    run<21>("4 exp + 12 fma mix", out, mhz);
    run<23>("v_fmac_f32", out, mhz); run<24>("v_add_f32", out, mhz); run<25>("v_mul_f32", out, mhz); run<26>("v_fma_f32 3 vgpr src", out, mhz); run<27>("v_fma_f32 sgpr src", out, mhz);
    run<0>("v_fma_f32", out, mhz); run<20>("v_xor_b32", out, mhz); run<3>("v_exp_f32", out, mhz); run<2>("v_mad_u64_u32 vcc", out, mhz);
    return 0;
}

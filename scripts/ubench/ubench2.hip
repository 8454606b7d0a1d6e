// operand-kind effects on VALU issue cost, gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 65536
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP> __global__ __launch_bounds__(1024) void k(float* out, int iters, float seed, float sc, unsigned m) {
    float a[8]; unsigned u[8]; float2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; u[i] = (unsigned)(threadIdx.x * 2654435761u + i); p[i] = make_float2(a[i], a[i] + 1); }
    const float c1 = 0.999f + seed * 1e-9f, c2 = 1e-3f + seed * 1e-9f;
    float2 sp = make_float2(sc, sc);
    for (int it = 0; it < iters; ++it) {
#define A0(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define A1(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(c2));
#define A2(i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(sc), "v"(c2));
#define A3(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "s"(sc));
#define A4(i) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a[i]) : "v"(c2));
#define A5(i) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define A6(i) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define A7(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));
#define A8(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(c2));
#define A9(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));
#define A10(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define A11(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define A12(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c2));
#define A13(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i+1)&7]));
#define A14(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(sp), "v"(p[(i+2)&7]));
#define A15(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "v"(u[(i+1)&7]), "v"(u[(i+2)&7]));
#define A16(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p[i]) : "v"(u[i]), "v"(u[(i+1)&7]) : "vcc");
#define A17(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define A18(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
#define A19(i) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(u[i]));
#define A20(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i+1)&7]));
#define A21(i) asm volatile("v_fma_f32 %0, %0, %1, %2 mul:2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define A22(i) asm volatile("v_exp_f32 %0, %0 mul:2" : "+v"(a[i]));
#define A23(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
#define A24(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
#define A25(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define A26(i) asm volatile("v_mul_f32 %0, %0, %1\n v_exp_f32 %0, %0" : "+v"(a[i]) : "v"(c1));
#define A27(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "s"(sc));
#define A28(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define A29(i) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(c2));
#define A30(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
#define A31(i) asm volatile("v_max_f32 %0, 0, %0" : "+v"(a[i]));
#define RUN(N) if (OP == N) { REP8(A##N) REP8(A##N) }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
        RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28) RUN(29) RUN(30) RUN(31)
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int OP> void run(const char* name, float* out, int n = 16) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, 1024>>>(out, ITERS, 1.0f, 0.5f, 0xD2511F53u);
    hipEventRecord(e0);
    k<OP><<<256, 1024>>>(out, ITERS, 1.0f, 0.5f, 0xD2511F53u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %6.2f cycles/instr/wave @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (4.0 * ITERS * n));
}
int main() {
    float* out; hipMalloc(&out, 64);
    run<0>("warm", out); run<0>("warm", out); run<0>("warm", out);
    run<0>("v_fma_f32 v,v,v", out); run<1>("v_fma_f32 v,s,v", out); run<2>("v_fma_f32 s,v,v", out); run<3>("v_fma_f32 v,v,s", out);
    run<27>("v_fma_f32 v,s,s(same)", out);
    run<4>("v_fma_f32 v,2.0,v", out); run<5>("v_fma_f32 -v,v,v", out); run<6>("v_fma_f32 |v|,v,v", out); run<21>("v_fma_f32 mul:2", out);
    run<7>("v_mul_f32 s,v (VOP2)", out); run<8>("v_fmac_f32 s,v (VOP2)", out); run<9>("v_add_f32 s,v (VOP2)", out);
    run<28>("v_add_f32_e64 v,v (VOP3)", out); run<29>("v_add_f32_e64 |v|,v", out);
    run<10>("v_max_f32", out); run<31>("v_max_f32 0,v", out); run<11>("v_sub_f32", out); run<12>("v_cndmask_b32 vcc", out); run<13>("v_mov_b32", out);
    run<14>("v_pk_fma_f32 v,s,v", out); run<15>("v_bitop3_b32 v,v,v", out); run<16>("v_mad_u64_u32 v,v", out); run<17>("v_med3_f32 v,v,v", out);
    run<18>("v_add_u32", out); run<19>("v_lshlrev_b32", out); run<20>("v_and_b32", out); run<22>("v_exp_f32 mul:2", out);
    run<23>("v_cvt_pk_f16_f32", out); run<24>("v_cvt_f32_f16", out); run<25>("v_fma_mix_f32", out); run<26>("v_mul+v_exp pair (per pair/2)", out, 32);
    run<30>("v_readlane_b32", out);
    return 0;
}

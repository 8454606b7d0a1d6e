// qbold_dev.h -- device-side building blocks shared by the gfx950 kernels of libqbold_hip.so.
//
// Everything here is per-voxel (or per voxel-sample) float32 arithmetic restating the reference's
// Python (file:line cited per function, relative to the reference repository root).  The kernels
// that use these pieces live in signal_kernels.hip / elbo_kernels.hip / encoder_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define QB_MAX_T 64
#define QB_TAB_SEG 256        // cubic-Hermite segments of the F(x) table (4 KiB in LDS)
#define QB_NNODE 129          // Simpson nodes of signals.py:168
#define QB_WAVE 64

// Folded float32 constants of one context; passed BY VALUE as a kernel argument so every field is
// read with scalar loads and lives in SGPRs (wave-uniform).
struct QbDev {
    int T, se_idx, full_model, include_blood;
    int multi_norm, predict_log, use_student_t, tissue_mode;
#ifdef QBOLD_ABLATION
    int debug_skip;  // timing experiments only (-DQBOLD_ABLATION builds of scripts/): 1 = no encoder MFMA, 2 = no sampling
#endif
    float dw_coef;   // (4/3) pi gamma b0 dchi hct                      signals.py:144
    float dw_coef_nohct;  // the same without hct (variable_hct)        signals.py:64-70
    float e_te_r2t;  // exp(-te*r2t)                                    signals.py:172
    float r2t_te;    // -r2t*te                                         signals.py:204
    float m_bld_nb;  // m_bld * nb                                      signals.py:102-107
    float g0_c1;     // (4/45) hct (1-hct)                              signals.py:239
    float g0_c2;     // 4 pi b0 dchi                                    signals.py:239
    float half_g2;   // 0.5 gamma^2                                     signals.py:241
    float td2;       // td^2                                            signals.py:241
    float e_r2b_te;  // exp(-r2b*te)                                    signals.py:241
    float bw_coef;   // include_blood ? m_bld_nb : 1      (fast path: blood weight = bw_coef * dbv, no select)
    float bwe_coef;  // include_blood ? e_r2b_te : 0      (fast path: blood_w = bw * bwe_coef)
    float tab_inv_h; // segments per unit x
    float tab_xmax;  // table covers |x| <= tab_xmax
    float dF_node0;  // slope of Simpson node 0 in dF/dx (see tissue_F)
    float st_df, st_const;  // Student-t df and log-normaliser          model.py:558
    // fast (table) path: u_t = |tauh0 + t*tauh_step| * dw is the table coordinate of tau_t
    float tauh0, tauh_step;  // tau_start * tab_inv_h, tau_step * tab_inv_h
    float ngk_l2e;           // -log2(e) * 0.5 gamma^2 (4/45) hct (1-hct) (4 pi b0 dchi)^2 td^2
    // the per-draw factors of FwdFast as affine functions of sb = sigmoid(b), DBV = 0.2 sb + 0.001 (model.py:299-305):
    float nd_a, nd_b;        // -log2(e) DBV                       = nd_a sb + nd_b
    float tw_a, tw_b;        // (1 - bw_coef DBV) exp(-te r2t)     = tw_a sb + tw_b
    float bv_a, bv_b;        // bw_coef DBV bwe_coef               = bv_a sb + bv_b
    float taus[QB_MAX_T];
    float blood_B[QB_MAX_T];  // bracket of signals.py:242-247 per tau
};

// Work-skipping hooks of the timing experiments (MEASUREMENTS.md 4.4 / 4.7).  They exist only in builds made with
// -DQBOLD_ABLATION (scripts/dev/build_ablation.sh); in the library the driver, the tests and bench.py load they are
// the constant 0 and the guarded code is unconditional.
#ifdef QBOLD_ABLATION
#define QB_ABLATE(c, bit) ((c).debug_skip & (bit))
#define QB_ABLATE_MASK(c) ((c).debug_skip)
#else
#define QB_ABLATE(c, bit) 0
#define QB_ABLATE_MASK(c) 0
#endif
// A wave-uniform condition the compiler cannot fold (always true).  The fused kernel's phases sit behind such
// branches: as separate basic blocks the encoder's MFMA chains and the sampling loops are scheduled and
// register-allocated on their own.  Measured: with the phases in one block (the round-2 ablation branches compiled
// out and nothing in their place) vi_fwd_kernel<11, 2, 2> spills 372 bytes per lane and takes 0.67 instead of
// 0.51 ms per 1 M voxels.
__device__ __forceinline__ bool qb_phase_fence() {
    int one = 1;
    asm volatile("" : "+s"(one));
    return one != 0;
}

namespace qb {

// ---------------------------------------------------------------------------------------------
// small math
// ---------------------------------------------------------------------------------------------
#define QB_LOG2E 1.4426950408889634f
#define QB_LN2 0.6931471805599453f
__device__ __forceinline__ float rcpf_(float v) { return __builtin_amdgcn_rcpf(v); }      // 1 ulp
__device__ __forceinline__ float exp2f_(float v) { return __builtin_amdgcn_exp2f(v); }    // v_exp_f32
__device__ __forceinline__ float log2f_(float v) { return __builtin_amdgcn_logf(v); }     // v_log_f32
__device__ __forceinline__ float sigmoidf_(float v) { return rcpf_(1.0f + exp2f_(-QB_LOG2E * v)); }
__device__ __forceinline__ float clampf_(float v, float lo, float hi) {
    return fminf(fmaxf(v, lo), hi);
}
// tanh through one exp: tanh(p) = 1 - 2/(exp(2p)+1); |error| ~ 1e-7 abs, monotone, saturates.
__device__ __forceinline__ float tanhf_(float p) {
    float e = exp2f_((2.0f * QB_LOG2E) * p);
    return 1.0f - 2.0f * rcpf_(e + 1.0f);
}
// transform_std / transform_offdiag -- model.py:288-294 = logit_mvn.py:91-97
__device__ __forceinline__ float transform_std(float p) { return tanhf_(p) * 3.0f - 1.0f; }
__device__ __forceinline__ float transform_offdiag(float p) {
    return tanhf_(p) * 0.1353352832366127f;  // exp(-2)
}

#define QB_OEF_RANGE 0.8f
#define QB_MIN_OEF 0.04f
#define QB_DBV_RANGE 0.2f
#define QB_MIN_DBV 0.001f

// Posterior parameters of one voxel with the transforms applied once.
struct LogitMvn {
    float mu_o, mu_d;  // logit-space means
    float s_o, s_d;    // log std (transformed)
    float c;           // off-diagonal Cholesky term (transformed)
    float e_so, e_sd;  // exp(s)
    float i_so, i_sd;  // exp(-s)
    float i_bl;        // -exp(-s_o-s_d)*c                               model.py:434
};

__device__ __forceinline__ LogitMvn make_mvn(const float p[5]) {
    LogitMvn m;
    m.mu_o = p[0];
    m.mu_d = p[2];
    m.s_o = transform_std(p[1]);
    m.s_d = transform_std(p[3]);
    m.c = transform_offdiag(p[4]);
    m.e_so = __expf(m.s_o);
    m.e_sd = __expf(m.s_d);
    m.i_so = __expf(-m.s_o);
    m.i_sd = __expf(-m.s_d);
    m.i_bl = __expf(-m.s_o - m.s_d) * m.c * -1.0f;
    return m;
}

// ReparamTrickLayer.call -- model.py:25-31: logit-space sample (a, b) from normals (z0, z1).
__device__ __forceinline__ void reparam_logits(const LogitMvn& m, float z0, float z1, float& a,
                                               float& b) {
    a = m.mu_o + z0 * m.e_so;
    b = m.mu_d + z0 * m.c + z1 * m.e_sd;
}
// forward_transform -- model.py:299-305
__device__ __forceinline__ void forward_transform(float a, float b, float& oef, float& dbv) {
    oef = sigmoidf_(a) * QB_OEF_RANGE + QB_MIN_OEF;
    dbv = sigmoidf_(b) * QB_DBV_RANGE + QB_MIN_DBV;
}

// Observation-side quantities of logit_gaussian_mvg_log_prob that do not depend on the
// distribution parameters (model.py:393-398): clipped unit-interval values' logits and Jacobian.
struct LogitObs {
    float l0, l1;  // logit(x)
    float jac;     // sum log x + log(1-x)
};
__device__ __forceinline__ LogitObs make_obs(float oef, float dbv) {
    // backwards_transform, model.py:307-311 (divisions by 0.8 / 0.2 as multiplies by 1.25 / 5)
    float x0 = (oef - QB_MIN_OEF) * 1.25f;
    float x1 = (dbv - QB_MIN_DBV) * 5.0f;
    x0 = clampf_(x0, 1e-6f, 1.0f - 1e-6f);         // model.py:394-395
    x1 = clampf_(x1, 1e-6f, 1.0f - 1e-6f);
    float lx0 = __logf(x0), l1x0 = __logf(1.0f - x0);
    float lx1 = __logf(x1), l1x1 = __logf(1.0f - x1);
    LogitObs o;
    o.l0 = lx0 - l1x0;  // logit, model.py:10-12
    o.l1 = lx1 - l1x1;
    o.jac = (lx0 + l1x0) + (lx1 + l1x1);  // model.py:398
    return o;
}
// Negative log-density -- gaussian_nll_chol + Jacobian, model.py:385-398, :423-447.
__device__ __forceinline__ float nlogp(const LogitObs& o, const LogitMvn& m) {
    float r0 = o.l0 - m.mu_o, r1 = o.l1 - m.mu_d;
    float w0 = r0 * m.i_so;
    float w1 = r1 * m.i_sd + r0 * m.i_bl;
    float swr = w0 * w0 + w1 * w1;
    return 1.8378770664093453f + (m.s_o + m.s_d) + 0.5f * swr + o.jac;  // log(2 pi) + ...
}

// ---------------------------------------------------------------------------------------------
// Philox4x32 (Random123) and the normal stream.
//
// The stream is this library's own definition (the reference's tf.random.normal stream is irreproducible);
// the oracle restates it word for word (oracle/qbold_oracle.c, qbo_philox_normals).  Round 4:
//   draw i of (seed, voxel, stream) = word (i & 3) of Philox4x32-7(ctr = (voxel_lo, voxel_hi, i >> 2, stream), key = seed)
//   word w -> radius  r = sqrt(-2 ln u1),  u1 = ((w & 0xffff) + 0.5) 2^-16   (u1 in (0, 1): |z| <= 4.8549)
//             angle   theta = (w >> 9) 2^-23 revolutions: its sixteen leading bits are the word's high half, independent
//                     of the radius; the seven bits below them (w[15:9], shared with the radius) move the angle by less
//                     than 2^-16 of a revolution -- a dither under the lattice's own resolution.  In this form the angle
//                     is ONE instruction: v_alignbit_b32(127, w, 9) = 0x3f800000 | (w >> 9) is the float 1 + theta, and
//                     v_sin_f32 / v_cos_f32 take revolutions, for which 1 + theta is theta.
//             (z0, z1) = r (cos 2 pi theta, sin 2 pi theta)
// i.e. one Philox call serves FOUR draws (rounds 1-3: two, with 32-bit uniforms and ten rounds) and a draw costs
// 7 + 10 vector instructions instead of 20 + 11.  Seven rounds are the Crush-resistant minimum Random123 publishes for
// Philox4x32 (Salmon et al., SC'11, table 2; known-answer vectors for 7 rounds: tests/test_oracle.py); sixteen-bit
// radius and angle put the normals on a 65,536 x 65,536 polar lattice (radial step at the mode 2.6e-5), whose moments
// tests/test_oracle.py holds against the Gaussian's.
// ---------------------------------------------------------------------------------------------
template <int ROUNDS>
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per word instead of a mul_hi / mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        // hi ^ ctr ^ key as one v_bitop3_b32 (truth table 0x96 = three-input XOR; the compiler emits two
        // v_xor_b32 for a ^ b ^ c on gfx950)
        c = make_uint4(__builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c.y, k.x, 0x96), (uint32_t)p1,
                       __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k.y, 0x96), (uint32_t)p0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) { return philox4x32<10>(c, k); }
__device__ __forceinline__ uint4 philox4x32_7(uint4 c, uint2 k) { return philox4x32<7>(c, k); }

#define QB_Z_MAX 4.8549f   // sqrt(-2 ln 2^-17) = 4.85487: the largest |z| the stream can produce

// 1 + theta as a float in [1, 2): the word's top 23 bits as the mantissa
__device__ __forceinline__ float bm_angle(uint32_t w) {
    return __builtin_bit_cast(float, __builtin_amdgcn_alignbit(127u, w, 9u));
}
// one word -> the two normals of one draw
__device__ __forceinline__ void box_muller16(uint32_t w, float& z0, float& z1) {
    const float u1 = fmaf((float)(w & 0xffffu), 0x1p-16f, 0x1p-17f);   // exact in float32
    const float th = bm_angle(w);
    // r = sqrt(-2 ln u1) via v_log_f32 (log2) and v_sqrt_f32; v_sin/v_cos take revolutions
    const float r = __builtin_amdgcn_sqrtf((-2.0f * QB_LN2) * log2f_(u1));
    z0 = r * __builtin_amdgcn_cosf(th);
    z1 = r * __builtin_amdgcn_sinf(th);
}
#define QB_BM_K 1.1774100225154747f   // sqrt(2 ln 2): z = QB_BM_K sqrt(-log2 u1) (cos, sin)

enum { STREAM_LIK = 0, STREAM_KL = 1, STREAM_MOMENTS = 2, STREAM_R2P = 3 };

// The four draws of Philox call `quad` of a stream, served one at a time: the words stay words until a draw is taken
// (four live registers instead of eight normals, and the Box-Muller code sits once in a draw loop that is not unrolled).
struct DrawQuad {
    uint32_t w0, w1, w2, w3;
    __device__ __forceinline__ void load(uint64_t seed, uint64_t vox, uint32_t quad, uint32_t stream) {
        const uint4 o = philox4x32_7(make_uint4((uint32_t)vox, (uint32_t)(vox >> 32), quad, stream),
                                     make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        w0 = o.x; w1 = o.y; w2 = o.z; w3 = o.w;
    }
    __device__ __forceinline__ void next(float& z0, float& z1) {   // draws 4 quad, 4 quad + 1, ... in order
        box_muller16(w0, z0, z1);
        w0 = w1; w1 = w2; w2 = w3;
    }
};

// normals of draws (4 quad .. 4 quad + 3), UNSCALED: z[2 d], z[2 d + 1] = z / QB_BM_K for draw 4 quad + d (a caller that
// only gathers moments of the draws scales the moments, once, instead of every normal); the draws d >= cnt (beyond the
// number asked for) come out as exact zeros -- one select on the radius each
__device__ __forceinline__ void normals8_unscaled(uint64_t seed, uint64_t vox, uint32_t quad, uint32_t stream, int cnt,
                                                  float z[8]) {
    const uint4 o = philox4x32_7(make_uint4((uint32_t)vox, (uint32_t)(vox >> 32), quad, stream),
                                 make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float u1 = fmaf((float)(w[d] & 0xffffu), 0x1p-16f, 0x1p-17f);
        const float th = bm_angle(w[d]);
        float r = __builtin_amdgcn_sqrtf(-log2f_(u1));
        if (d > 0) r = d < cnt ? r : 0.0f;
        z[2 * d] = r * __builtin_amdgcn_cosf(th);
        z[2 * d + 1] = r * __builtin_amdgcn_sinf(th);
    }
}
__device__ __forceinline__ void normals8(uint64_t seed, uint64_t vox, uint32_t quad, uint32_t stream, int cnt, float z[8]) {
    normals8_unscaled(seed, vox, quad, stream, cnt, z);
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] *= QB_BM_K;
}

// normals of draws (2*pair, 2*pair+1) of `stream` for global voxel `vox`: half a Philox call's words (the kernels off
// the hot path walk their draws in pairs; the hot loops take whole calls through DrawQuad / normals8)
__device__ __forceinline__ void normals4(uint64_t seed, uint64_t vox, uint32_t pair,
                                         uint32_t stream, float z[4]) {
    const uint4 o = philox4x32_7(make_uint4((uint32_t)vox, (uint32_t)(vox >> 32), pair >> 1, stream),
                                 make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const bool hi = (pair & 1u) != 0u;
    box_muller16(hi ? o.z : o.x, z[0], z[1]);
    box_muller16(hi ? o.w : o.y, z[2], z[3]);
}

// ---------------------------------------------------------------------------------------------
// Tissue integral F(x) -- signals.py:159-185.
// ---------------------------------------------------------------------------------------------
// Cephes j0f (= Eigen generic_j0<float> behind tf.math.special.bessel_j0, signals.py:170).
__device__ __forceinline__ float j0f_(float xx) {
    float x = fabsf(xx);
    float z = x * x;
    float p = -6.068350350393235E-008f;
    p = p * z + 6.388945720783375E-006f;
    p = p * z + -3.969646342510940E-004f;
    p = p * z + 1.332913422519003E-002f;
    p = p * z + -1.729150680240724E-001f;
    float small = (x < 1.0e-3f) ? (1.0f - 0.25f * z) : (z - 5.78318596294678452118f) * p;
    if (x <= 2.0f) return small;
    float q = 1.0f / x;
    float w = rsqrtf(x);
    float m = -6.838999669318810E-002f;
    m = m * q + 1.864949361379502E-001f;
    m = m * q + -2.145007480346739E-001f;
    m = m * q + 1.197549369473540E-001f;
    m = m * q + -3.560281861530129E-003f;
    m = m * q + -4.969382655296620E-002f;
    m = m * q + -3.355424622293709E-006f;
    m = m * q + 7.978845717621440E-001f;
    float w2 = q * q;
    float h = 3.242077816988247E+001f;
    h = h * w2 + -3.630592630518434E+001f;
    h = h * w2 + 1.756221482109099E+001f;
    h = h * w2 + -4.974978466280903E+000f;
    h = h * w2 + 1.001973420681837E+000f;
    h = h * w2 + -1.939906941791308E-001f;
    h = h * w2 + 6.490598792654666E-002f;
    h = h * w2 + -1.249992184872738E-001f;
    float xn = q * h - 0.7853981633974483096f;
    return (w * m) * cosf(xn + x);
}
// Cephes j1f (the derivative TF registers for bessel_j0 is -bessel_j1).
__device__ __forceinline__ float j1f_(float xx) {
    float x = fabsf(xx);
    float r;
    if (x <= 2.0f) {
        float z = x * x;
        float p = -4.878788132172128E-009f;
        p = p * z + 6.009061827883699E-007f;
        p = p * z + -4.541343896997497E-005f;
        p = p * z + 1.937383947804541E-003f;
        p = p * z + -3.405537384615824E-002f;
        r = (z - 1.46819706421238932572E1f) * x * p;
    } else {
        float q = 1.0f / x;
        float w = sqrtf(q);
        float m = 6.913942741265801E-002f;
        m = m * q + -2.284801500053359E-001f;
        m = m * q + 3.138238455499697E-001f;
        m = m * q + -2.102302420403875E-001f;
        m = m * q + 5.435364690523026E-003f;
        m = m * q + 1.493389585089498E-001f;
        m = m * q + 4.976029650847191E-006f;
        m = m * q + 7.978845453073848E-001f;
        float w2 = q * q;
        float h = -4.497014141919556E+001f;
        h = h * w2 + 5.073465654089319E+001f;
        h = h * w2 + -2.485774108720340E+001f;
        h = h * w2 + 7.222973196770240E+000f;
        h = h * w2 + -1.544842782180211E+000f;
        h = h * w2 + 3.503787691653334E-001f;
        h = h * w2 + -1.637986776941202E-001f;
        h = h * w2 + 3.749989509080821E-001f;
        float xn = q * h - 2.35619449019234492885f;
        r = (w * m) * cosf(xn + x);
    }
    return xx < 0 ? -r : r;
}

// LDS image shared by every kernel that evaluates the forward model:
//   tab[QB_TAB_SEG] float4 cubic coefficients, then the Simpson node tables (literal mode).
struct FwdLds {
    float4 tab[QB_TAB_SEG];
    float u[QB_NNODE + 3];
    float pre[QB_NNODE + 3];  // (2+u)*sqrt(1-u)
    float den[QB_NNODE + 3];  // 3*u^2
    float blood_B[QB_MAX_T];  // QbDev::blood_B for the fused forward kernel's draw loop (a VGPR operand
                              // instead of an SGPR one: v_fma_f32 issues in 2.9 instead of 5.5 cycles)
};

// ---- per-tau tissue table of the sampling fast path ------------------------------------------------------------
// Inside the ELBO kernels every (OEF, DBV) comes from forward_transform, so 0.04 <= OEF <= 0.84, and the protocol's
// taus are j * |tau_step| about the spin echo (tau = 0 there, compile-time index SE).  F(|tau_j| dw(OEF)) is then
// tabulated per tau ON ONE OEF GRID: G_j(OEF), cubic-Hermite segments in OEF.  The segment index and the fraction
// are computed ONCE PER DRAW and serve every tau -- the per-tau coordinate FMA, v_cvt_i32, v_fract and address
// shift of the x-indexed table are gone (3.5 of ~13 vector instructions per (draw, tau)), and the rows of all taus
// sit at immediate offsets from one address register.  Accuracy (float64 construction, float32 coefficients): max
// |G - F| = 4.7e-7 at 138 segments (T = 11), below the x-indexed table's own 9e-7 (whose float32 coordinate
// arithmetic this form avoids).
__host__ __device__ constexpr int gtab_taus(int T, int SE) { return SE > T - 1 - SE ? SE : T - 1 - SE; }
#ifndef QB_GT_SEGS_11
#define QB_GT_SEGS_11 138
#endif
// T = 24: 16 taus x 52 segments = 13 KiB is what the LDS has left beside that protocol's larger weight image (max
// |G - F| = 2.3e-5 at 64 ms, i.e. <= 4.7e-6 of the signal at DBV = 0.2).  Round 3 measured this form slower (0.947
// against 0.755 ms): with 48 data registers live its row ring spilled; with the mirrored pairs merged (34 data registers,
// elbo_core.h) it fits.
#ifndef QB_GT_SEGS_24
#define QB_GT_SEGS_24 52
#endif
// T = 64 (BASELINE config 3, spin echo at index 12): 51 taus x 68 segments = 54 KiB beside the 104 KiB of a 256-voxel
// tile's data rows (elbo_fwd_gt64_kernel); max |G - F| = 8e-6 at 65 ms, i.e. <= 1.6e-6 of the signal at DBV = 0.2.
#ifndef QB_GT_SEGS_64
#define QB_GT_SEGS_64 68
#endif
__host__ __device__ constexpr int gtab_segs(int T) {
    return T == 11 ? QB_GT_SEGS_11 : (T == 24 ? QB_GT_SEGS_24 : (T == 64 ? QB_GT_SEGS_64 : 0));
}
// Pair-interleaved rows (QB_GT_PAIRS): when every evaluated tau lies above the spin echo and they come in pairs, rows
// 2p and 2p + 1 of the table hold (c0a, c0b, c1a, c1b) and (c2a, c2b, c3a, c3b) of taus j = 2p + 1 (a) and 2p + 2 (b),
// so that a row read delivers aligned float32 pairs for v_pk_fma_f32 (elbo_core.h).  Measured and NOT adopted: 137
// instead of 157 instructions per draw, but 0.474 against 0.459 ms per 1 M voxels on the split-f16 kernel (the packed
// forms issue slower than they save, as round 1 found for hand-built pairs); kept as a build option.
#ifndef QB_GT_PAIRS
#define QB_GT_PAIRS 0
#endif
__host__ __device__ constexpr bool gtab_paired(int T, int SE) {
    return QB_GT_PAIRS && gtab_segs(T) > 0 && 2 * SE - (T - 1) <= 0 && (T - 1 - SE) % 2 == 0 && gtab_taus(T, SE) == T - 1 - SE;
}
#define QB_GT_OEF_MIN 0.04f
#define QB_GT_OEF_RANGE 0.8f
template <int T, int SE>
struct GtLds {
    static constexpr int J = gtab_taus(T, SE), NSEG = gtab_segs(T);
    float4 gtab[J * NSEG > 0 ? J * NSEG : 1];        // [j - 1][segment]: G_j on segment i as c0 + f (c1 + f (c2 + f c3))
    float blood_B[(T + 3) & ~3];  // QbDev::blood_B (a VGPR operand instead of an SGPR one)
};
template <int T, int SE>
__device__ __forceinline__ void gt_lds_fill(GtLds<T, SE>* L, const float4* __restrict__ g_gtab, const QbDev& c) {
    for (int i = threadIdx.x; i < GtLds<T, SE>::J * GtLds<T, SE>::NSEG; i += blockDim.x) L->gtab[i] = g_gtab[i];
    if (threadIdx.x < T) L->blood_B[threadIdx.x] = c.blood_B[threadIdx.x];
}

__device__ __forceinline__ void fwd_lds_fill(FwdLds* L, const float4* __restrict__ g_tab,
                                             bool literal) {
    for (int i = threadIdx.x; i < QB_TAB_SEG; i += blockDim.x) L->tab[i] = g_tab[i];
    if (literal) {
        // tf.linspace(1e-5, 1, 129) in float32 -- signals.py:166-168
        const float a = 1e-5f, b = 1.0f;
        const float delta = (b - a) / 128.0f;
        for (int i = threadIdx.x; i < QB_NNODE; i += blockDim.x) {
            float u = (i == QB_NNODE - 1) ? b : a + (float)i * delta;
            L->u[i] = u;
            L->pre[i] = (2.0f + u) * sqrtf(1.0f - u);
            L->den[i] = 3.0f * (u * u);
        }
    }
}

// 129-node Simpson sum in float32, the reference's operation order -- signals.py:169-185.
__device__ __noinline__ float tissue_F_literal(const FwdLds* L, float x) {
    const float h3 = ((L->u[2] - L->u[0]) / 2.0f) / 3.0f;
    const float x15 = 1.5f * x;
    float acc = 0.0f;
    float ya = L->pre[0] * (1.0f - j0f_(x15 * L->u[0])) / L->den[0];
    for (int m = 0; m < (QB_NNODE - 1) / 2; ++m) {
        float ym = L->pre[2 * m + 1] * (1.0f - j0f_(x15 * L->u[2 * m + 1])) / L->den[2 * m + 1];
        float yb = L->pre[2 * m + 2] * (1.0f - j0f_(x15 * L->u[2 * m + 2])) / L->den[2 * m + 2];
        acc += (ya + yb + 4.0f * ym) * h3;
        ya = yb;
    }
    return acc;
}
__device__ __noinline__ float tissue_dF_literal(const FwdLds* L, float x) {
    const float h3 = ((L->u[2] - L->u[0]) / 2.0f) / 3.0f;
    const float x15 = 1.5f * x;
    float acc = 0.0f;
    float ya = L->pre[0] * (1.5f * L->u[0] * j1f_(x15 * L->u[0])) / L->den[0];
    for (int m = 0; m < (QB_NNODE - 1) / 2; ++m) {
        int i1 = 2 * m + 1, i2 = 2 * m + 2;
        float ym = L->pre[i1] * (1.5f * L->u[i1] * j1f_(x15 * L->u[i1])) / L->den[i1];
        float yb = L->pre[i2] * (1.5f * L->u[i2] * j1f_(x15 * L->u[i2])) / L->den[i2];
        acc += (ya + yb + 4.0f * ym) * h3;
        ya = yb;
    }
    return acc;
}

// Table evaluation: F and (optionally) dF/dx.  |x| beyond the table (OEF > 1, never produced by
// forward_transform) takes the literal sum.  The derivative is the one TensorFlow's autodiff
// yields through bessel_j0 (its registered gradient is -J1): the J1-kernel Simpson sum over ALL
// 129 nodes.  Node 0 (u = 1e-5) contributes 0 to the float32 forward value (1 - J0 rounds to 0)
// but a slope of dF_node0 * x to that gradient (J1(z) ~ z/2 is representable), so it is added to
// the derivative of the forward cubic.
template <bool LITERAL, bool WITH_D>
__device__ __forceinline__ float tissue_F(const FwdLds* L, const QbDev& c, float x, float* dF) {
    float ax = fabsf(x);
    if (LITERAL || ax > c.tab_xmax) {
        if (WITH_D) {
            float d = tissue_dF_literal(L, ax);
            *dF = x < 0 ? -d : d;
        }
        return tissue_F_literal(L, ax);
    }
    float u = ax * c.tab_inv_h;
    int i = min((int)u, QB_TAB_SEG - 1);
    float f = u - (float)i;
    float4 k = L->tab[i];
    if (WITH_D) {
        float d = fmaf(fmaf(3.0f * k.w, f, 2.0f * k.z), f, k.y) * c.tab_inv_h + c.dF_node0 * ax;
        *dF = x < 0 ? -d : d;
    }
    return fmaf(fmaf(fmaf(k.w, f, k.z), f, k.y), f, k.x);
}

// Per-voxel-sample part of the forward model that does not depend on tau.
struct FwdVox {
    float dw;     // calculate_dw, signals.py:142-147
    float dbv;
    float bw;     // blood weight, signals.py:107 / :110
    float tw;     // tissue weight, signals.py:112
    float g;      // 0.5 gamma^2 g0 td^2, signals.py:239-241
};
__device__ __forceinline__ FwdVox fwd_vox(const QbDev& c, float oef, float dbv) {
    FwdVox v;
    v.dw = c.dw_coef * oef;
    v.dbv = dbv;
    if (c.include_blood) {
        v.bw = c.m_bld_nb * dbv;
        float t = c.g0_c2 * oef;
        v.g = (c.half_g2 * (c.g0_c1 * (t * t))) * c.td2;
    } else {
        v.bw = dbv;
        v.g = 0.0f;
    }
    v.tw = 1.0f - v.bw;
    return v;
}
// variable_hct (signals.py:64-70): hct is a float32 tensor, so the float64 constants round before it
// multiplies: ((K * hct) * oef) and ((4/45 * hct) * (1 - hct)).
__device__ __forceinline__ FwdVox fwd_vox_hct(const QbDev& c, float oef, float dbv, float hct) {
    FwdVox v;
    v.dw = (c.dw_coef_nohct * hct) * oef;
    v.dbv = dbv;
    if (c.include_blood) {
        v.bw = c.m_bld_nb * dbv;
        float t = c.g0_c2 * oef;
        v.g = (c.half_g2 * (((float)(4.0 / 45.0) * hct) * (1.0f - hct) * (t * t))) * c.td2;
    } else {
        v.bw = dbv;
        v.g = 0.0f;
    }
    v.tw = 1.0f - v.bw;
    return v;
}
// signal at one tau -- signals.py:98-114 with calc_tissue :152-209 and calc_blood :233-247.
template <bool LITERAL>
__device__ __forceinline__ float fwd_tissue(const FwdLds* L, const QbDev& c, const FwdVox& v, int t) {
    const float tau = c.taus[t];
    if (c.full_model) {
        // tau = 0 (the spin echo): every node's 1 - j0f(0) is exactly 0 and the table's first coefficient is
        // F(0) = 0, so F = 0 and exp(-dbv * 0) = 1 in either mode -- same float32 result without the sum
        if (tau == 0.0f) return c.e_te_r2t;
        float F = tissue_F<LITERAL, false>(L, c, tau * v.dw, nullptr);
        return __expf(-v.dbv * F) * c.e_te_r2t;
    }
    // log-linear two-regime model, signals.py:194-207
    float tc = 1.0f / v.dw;
    float rt = (v.dw * v.dbv) * tau;
    float e = __expf(c.r2t_te);
    return (fabsf(tau) < tc) ? e * __expf(-(0.3f * (rt * rt)) / v.dbv) : e * __expf(v.dbv - rt);
}
__device__ __forceinline__ float fwd_mix(const QbDev& c, const FwdVox& v, float tissue, int t) {
    float blood = c.include_blood ? c.e_r2b_te * __expf(-v.g * c.blood_B[t]) : 0.0f;
    return v.tw * tissue + v.bw * blood;
}
template <bool LITERAL>
__device__ __forceinline__ float fwd_signal(const FwdLds* L, const QbDev& c, const FwdVox& v,
                                            int t) {
    return fwd_mix(c, v, fwd_tissue<LITERAL>(L, c, v, t), t);
}

// Fast path of the same model (full model, table mode, inputs from forward_transform so that
// 0.04 <= OEF <= 0.84 and 0.001 <= DBV <= 0.201): the table coordinate is a single FMA per tau,
// constants are folded once per draw, exponentials go straight to v_exp_f32.
struct FwdFast {
    float ua, ub;   // u_t = |ua + t * ub|
    float nd;       // -dbv * log2(e)
    float tissue_w; // (1 - bw) * exp(-te r2t)
    float blood_w;  // bw * exp(-r2b te)           (0 without blood)
    float ng;       // -g * log2(e)
};
__device__ __forceinline__ FwdFast fwd_fast(const QbDev& c, float oef, float dbv) {
    FwdFast v;
    const float dw = c.dw_coef * oef;
    v.ua = c.tauh0 * dw;
    v.ub = c.tauh_step * dw;
    v.nd = -QB_LOG2E * dbv;
    const float bw = c.bw_coef * dbv;  // m_bld_nb * dbv, or dbv without the blood compartment
    v.tissue_w = (1.0f - bw) * c.e_te_r2t;
    v.blood_w = bw * c.bwe_coef;       // bw * exp(-r2b te), or 0
    v.ng = c.ngk_l2e * (oef * oef);
    return v;
}
// The same from the two sigmoids of a draw (sa, sb: OEF = 0.8 sa + 0.04, DBV = 0.2 sb + 0.001): every factor that is
// affine in DBV is one FMA on sb with constants folded on the host (five instructions instead of eight; the table
// coordinates ua / ub of the x-indexed table are not formed -- for the per-tau table, whose coordinate is sa itself).
__device__ __forceinline__ FwdFast fwd_fast_sig(const QbDev& c, float sa, float sb) {
    FwdFast v;
    v.ua = v.ub = 0.0f;
    v.nd = fmaf(sb, c.nd_a, c.nd_b);
    v.tissue_w = fmaf(sb, c.tw_a, c.tw_b);
    v.blood_w = fmaf(sb, c.bv_a, c.bv_b);
    const float oef = fmaf(sa, QB_OEF_RANGE, QB_MIN_OEF);
    v.ng = c.ngk_l2e * (oef * oef);
    return v;
}
__device__ __forceinline__ float fwd_signal_fast(const FwdLds* L, const QbDev& c, const FwdFast& v,
                                                 int t) {
    // 0 <= u < QB_TAB_SEG * 0.84: forward_transform bounds OEF by 0.84 and the table spans OEF <= 1,
    // so the segment index needs no clamp (an out-of-range LDS read would return 0, not fault)
    const float u = fabsf(fmaf((float)t, v.ub, v.ua));
    const int i = (int)u;
    const float f = __builtin_amdgcn_fractf(u);
    const float4 k = L->tab[i];
    const float F = fmaf(fmaf(fmaf(k.w, f, k.z), f, k.y), f, k.x);
    const float e1 = exp2f_(v.nd * F);
    const float e2 = exp2f_(v.ng * c.blood_B[t]);
    return fmaf(v.tissue_w, e1, v.blood_w * e2);
}

// ---------------------------------------------------------------------------------------------
// wave / block reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace qb

// elbo_core.h -- per-voxel Monte-Carlo ELBO terms, shared by the stand-alone ELBO kernel
// (elbo_kernels.hip) and the fused encoder+ELBO kernel (vi_kernels.hip).
//
// A wave owns 16 voxels; the four lanes l, l+16, l+32, l+48 share voxel (l & 15) and split its
// Monte-Carlo draws by Philox call (call g -> draws 4g .. 4g+3, qbold_dev.h; call g belongs to lane group
// g & 3).  Everything a lane needs per voxel lives in registers: T normalised data points, T
// inverse sigmas, the transformed posterior / prior parameters.
#pragma once
// Wave priorities of the sampling phases (s_setprio; the fused kernel runs its encoder phase at 2-3, see
// vi_kernels.hip).  Measured on the fused kernel at sustained clocks, 1 M voxels: all phases equal 0.597 ms;
// likelihood loop 1 / KL loop 3 / everything after the draws 0: 0.572 ms.  Preferring the waves that are
// about to finish a tile (KL) and those that feed the matrix pipe over the long likelihood loops keeps the
// four waves of a SIMD in different phases.
#ifndef QB_PRIO_LIK
#define QB_PRIO_LIK 1
#endif
#ifndef QB_LIK_BARRIER
#define QB_LIK_BARRIER 2  // compiler barrier every this many evaluated taus of the likelihood loop (1: 0.517,
                          // 2: 0.509, 3: 0.515, 4: 0.512, 8: 0.512, none: 0.514 ms)
#endif
#ifndef QB_GT_DEPTH
#define QB_GT_DEPTH 8   // table rows requested ahead in the per-tau-table likelihood loop (measured on the fused
                        // kernel, 1 M voxels: 2: 0.4630, 4: 0.4608, 8 = all of a draw's rows: 0.4596 ms)
#endif
#ifndef QB_GT_DEPTH_LONG
#define QB_GT_DEPTH_LONG 2   // the same for protocols of more than 16 taus (T = 24: 34 data registers live; scratch per lane
                             // 296 / 196 / 144 bytes at depth 8 / 4 / 2 before the rest of the kernel was trimmed)
#endif
#ifndef QB_X_DEPTH
#define QB_X_DEPTH 2    // the same ring on the x-indexed table (compile-time spin-echo kernels without a per-tau table)
#endif
#ifndef QB_PRIO_KL
#define QB_PRIO_KL 2    // round 3, whitened KL loop: 3: 0.4630, 2: 0.4607, 1: 0.4642 ms
#endif
#ifndef QB_PRIO_AFTER
#define QB_PRIO_AFTER 0
#endif

#include "qbold_dev.h"

namespace qb {

#define QB_LANES_PER_VOXEL 4
#define QB_VOX_PER_WAVE 16

// sum over the four lanes of a voxel (lane groups 16 apart)
__device__ __forceinline__ float voxel_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Likelihood side of one voxel, prepared once -- fine_tune_loss_fn, model.py:527-568.
template <int T>
struct VoxelLik {
    float yt[T];      // normalised (optionally log) data                model.py:541-549
    float inv_s[T];   // 1/sigma
    float log_s_sum;  // sum_t log sigma_t (+ T log sqrt(2 pi) for the Gaussian)
    float mask;
};

// Normaliser of model.py:541-545: v[se] (or the mean of v[se-1..se+1]) + 1e-3.  SE >= 0 means
// "single-image normalisation with compile-time spin-echo index SE" (the host dispatch selects
// it only when !multi_image_normalisation); SE < 0 reads c.se_idx / c.multi_norm at run time
// without dynamic register indexing.
template <int T, int SE>
__device__ __forceinline__ float se_norm(const QbDev& c, const float (&v)[T]) {
    if (SE >= 0) return v[SE >= 0 ? SE : 0] + 1e-3f;
    const int se = c.se_idx;
    float a = 0.0f, b = 0.0f, d = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        a = (t == se - 1) ? v[t] : a;
        b = (t == se) ? v[t] : b;
        d = (t == se + 1) ? v[t] : d;
    }
    return c.multi_norm ? (a + b + d) / 3.0f + 1e-3f : b + 1e-3f;
}

// LOGSIG: `sigma` holds log sigma (the encoder's pre-activation, model.py:211-214) instead of sigma.
// PRESCALE (fast path with a compile-time spin-echo index only): yt holds yt / sigma, see sample_sq_fast.
// LINEAR: the caller dispatched on the fast path (Gaussian likelihood on linear data), so the log-data and
// Student-t switches are compiled out (left as run-time selects they cost 11 v_log + 50 selects per tile).
// MIR (with PRESCALE; the protocol mirrors about its spin-echo image, qbold_ctx::grid_mirrors): the signal is even in
// tau, so tau_{SE+j} and tau_{SE-j} share ONE prediction yh, and their two residuals are one:
//   (a1 - yh s1)^2 + (a2 - yh s2)^2 = (Q - yh P)^2 + D,   P = sqrt(s1^2 + s2^2),  Q = (a1 s1 + a2 s2) / P,
//   D = ((a1 s2 - a2 s1) / P)^2                      (a = y / sigma, s = 1 / sigma; exact algebra, no cancellation)
// The pair is stored as the single data point (Q, P) at index SE + j, its draw-independent remainder D goes into the
// per-draw constant, and the entries below the spin echo are dead: two registers and two FMAs less per pair and draw
// (T = 11: 2 pairs, T = 24: 7 -- the 24-tau kernel's spills -- T = 64: 12).
template <int T, int SE, bool LOGSIG, bool PRESCALE = false, bool LINEAR = false, bool MIR = false>
__device__ __forceinline__ void prepare_lik(const QbDev& c, const float (&x)[T],
                                            const float (&sigma)[T], float mask, VoxelLik<T>& k) {
    static_assert(!MIR || (PRESCALE && SE >= 0), "merged mirror pairs need pre-scaled data and a compile-time spin echo");
    const float inv_nt = rcpf_(se_norm<T, SE>(c, x));
    float ls = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float y = x[t] * inv_nt;
        if (!LINEAR && c.predict_log) y = mask > 0.0f ? __logf(y) : 0.0f;  // model.py:548
        if (LOGSIG) {
            k.inv_s[t] = exp2f_(-QB_LOG2E * sigma[t]);
            ls += sigma[t];
        } else {
            k.inv_s[t] = rcpf_(sigma[t]);
            ls += QB_LN2 * log2f_(sigma[t]);
        }
        k.yt[t] = y;
    }
    if (PRESCALE) {
#pragma unroll
        for (int t = 0; t < T; ++t) k.yt[t] *= k.inv_s[t];
    }
    float half_d = 0.0f;
    if constexpr (MIR) {
        constexpr int kSE = SE >= 0 ? SE : 0;
        float dsum = 0.0f;
#pragma unroll
        for (int t = kSE + 1; t < T; ++t) {
            const int tm = 2 * kSE - t;
            if (tm >= 0) {
                const float a1 = k.yt[t], s1 = k.inv_s[t], a2 = k.yt[tm >= 0 ? tm : 0], s2 = k.inv_s[tm >= 0 ? tm : 0];
                const float p2 = fmaf(s1, s1, s2 * s2);
                const float ip = __builtin_amdgcn_rsqf(p2);
                const float cr = fmaf(a1, s2, -(a2 * s1)) * ip;
                k.inv_s[t] = p2 * ip;
                k.yt[t] = fmaf(a1, s1, a2 * s2) * ip;
                dsum = fmaf(cr, cr, dsum);
            }
        }
        half_d = 0.5f * dsum;
    }
    k.log_s_sum = ((!LINEAR && c.use_student_t) ? ls : ls + (float)T * 0.9189385332046727f) + half_d;  // log sqrt(2 pi)
    k.mask = mask;
}

// NLL of one reparameterised draw: forward model over the T taus, normalise, score.
template <int T, int SE, bool LITERAL>
__device__ __forceinline__ float sample_nll(const FwdLds* L, const QbDev& c, const VoxelLik<T>& k,
                                            float oef, float dbv) {
    const FwdVox fv = fwd_vox(c, oef, dbv);
    float s[T];
    if (SE >= 0) {
        // Compile-time spin-echo index: the full model's tissue factor is even in tau (F takes |x|; the
        // literal Simpson sum sees tau only through j0f(|.|)), so a pair of taus that mirror EXACTLY in
        // float32 about the spin echo shares one tissue factor -- the same float32 value either way.  (The
        // reference's float32 grid start + i step mirrors only in part: -16 / +16 ms do, -8 / +8 ms differ
        // by an ulp and are evaluated separately.)  The blood factor keeps its own bracket per tau.
        constexpr int kSE = SE >= 0 ? SE : 0;
        float tis[T];
#pragma unroll
        for (int t = T - 1; t >= 0; --t) {
            const int m = (t < kSE && 2 * kSE - t < T) ? 2 * kSE - t : 0;
            const bool mirrored = t < kSE && 2 * kSE - t < T && c.full_model && c.taus[m] == -c.taus[t];
            if (mirrored) tis[t] = tis[m];
            else tis[t] = fwd_tissue<LITERAL>(L, c, fv, t);
            s[t] = fwd_mix(c, fv, tis[t], t);
        }
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) s[t] = fwd_signal<LITERAL>(L, c, fv, t);
    }
    const float inv_np = 1.0f / se_norm<T, SE>(c, s);  // model.py:542 / :545
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float yp = s[t] * inv_np;
        if (c.predict_log) yp = k.mask > 0.0f ? __logf(yp) : 0.0f;  // model.py:549
        float r = (k.yt[t] - yp) * k.inv_s[t];                      // model.py:552, 561
        if (c.use_student_t)  // -StudentT(df, 0, sigma).log_prob(res), model.py:558-559
            acc += 0.5f * (c.st_df + 1.0f) * log1pf(r * r / c.st_df) - c.st_const;
        else
            acc = fmaf(0.5f * r, r, acc);
    }
    return acc + k.log_s_sum;  // model.py:563
}

// Fast path (full model, table mode, Gaussian likelihood on linear data): same arithmetic with the
// per-draw constants folded (FwdFast) and the residual scored as sum r^2 (0.5 applied once).
template <class LDS>
struct IsGtLds { static constexpr bool value = false; };
template <int T, int SE>
struct IsGtLds<GtLds<T, SE>> { static constexpr bool value = true; };

// The same with the per-tau OEF-indexed table (GtLds; qbold_dev.h): the host dispatches here only for protocols with
// tau = 0 at the compile-time spin-echo index and a table built for them (qbold_ctx::gtab_ok).
// sa, sb: the draw's two sigmoids (OEF = 0.8 sa + 0.04, DBV = 0.2 sb + 0.001, model.py:299-305).
template <int T, int SE>
__device__ __forceinline__ float sample_sq_fast(const GtLds<T, SE>* L, const QbDev& c, const VoxelLik<T>& k,
                                                float sa, float sb) {
    static_assert(SE >= 0 && gtab_segs(T) > 0, "built for compile-time spin-echo protocols");
    constexpr int NSEG = gtab_segs(T);
    const FwdFast fv = fwd_fast_sig(c, sa, sb);
    float acc = 0.0f;
    const float s_se = fmaf(fv.tissue_w, 1.0f, fv.blood_w * exp2f_(fv.ng * L->blood_B[SE]));   // F(0) = 0
    const float inv_np = rcpf_(s_se + 1e-3f);
    const float lt = log2f_(fv.tissue_w * inv_np), lb = log2f_(fv.blood_w * inv_np);
    // one coordinate per draw: OEF in [0.04, 0.84] -> segment index and fraction shared by every tau.  The table's
    // grid is OEF = 0.04 + 0.8 i / NSEG, so the coordinate is NSEG sa (sa in [0, 1]; 1 only by rounding: clamped)
    static_assert(QB_GT_OEF_MIN == QB_MIN_OEF && QB_GT_OEF_RANGE == QB_OEF_RANGE, "the table spans forward_transform's range");
    const float cg = fminf(sa * (float)NSEG, (float)NSEG - 0.0009765625f);
    const float f = __builtin_amdgcn_fractf(cg);
    const float4* row = L->gtab + (int)cg;
    {
        const float r = fmaf(-(s_se * inv_np), k.inv_s[SE], k.yt[SE]);
        acc = fmaf(r, r, acc);
    }
    // evaluation order: SE + 1 .. T - 1, then the taus below the spin echo that have no partner on the grid
    constexpr int NA = T - 1 - SE, NB = (2 * SE - (T - 1)) > 0 ? 2 * SE - (T - 1) : 0, NE = NA + NB;
    auto score = [&](int t, float yh) {   // the normalised prediction at tau index t against the data point at t: for
        const float r = fmaf(-yh, k.inv_s[t], k.yt[t]);   // t > SE with a mirror on the grid that is the MERGED pair
        acc = fmaf(r, r, acc);                            // (prepare_lik<.., MIR>), the signal being even in tau
    };
    // The rows of a draw sit at immediate offsets from one address, so nothing orders their reads: left alone the
    // compiler requests all of them up front and spills under the fused kernel's 128-register budget.  A ring keeps
    // QB_GT_DEPTH evaluations in flight; sched_barrier pins requests AND arithmetic.
    if constexpr (gtab_paired(T, SE)) {
        // Two taus per step on packed float32 pairs (v_pk_fma_f32: 4.6 cycles for two FMAs against 2 x 2.9): the table
        // interleaves the coefficients of taus 2p + 1 and 2p + 2 so that a row read delivers aligned pairs -- cubic,
        // both exponent FMAs and the blood bracket run packed, no register moves to build the pairs.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        static_assert(NB == 0 && NA % 2 == 0, "paired table: every evaluated tau above the spin echo, in pairs");
        struct Stage {
            float4 A, B;   // (c0a, c0b, c1a, c1b), (c2a, c2b, c3a, c3b)
            f32x2 bb;
        };
        auto issue = [&](int p) -> Stage {
            Stage st;
            st.A = row[(2 * p) * NSEG];
            st.B = row[(2 * p + 1) * NSEG];
            st.bb = f32x2{L->blood_B[SE + 1 + 2 * p], L->blood_B[SE + 2 + 2 * p]};
            return st;
        };
        const f32x2 ff{f, f}, nd2{fv.nd, fv.nd}, ng2{fv.ng, fv.ng}, lt2{lt, lt}, lb2{lb, lb};
        auto finish = [&](int p, const Stage& st) {
            const f32x2 c0{st.A.x, st.A.y}, c1{st.A.z, st.A.w}, c2{st.B.x, st.B.y}, c3{st.B.z, st.B.w};
            const f32x2 F = __builtin_elementwise_fma(__builtin_elementwise_fma(__builtin_elementwise_fma(c3, ff, c2), ff, c1), ff, c0);
            const f32x2 e1 = __builtin_elementwise_fma(nd2, F, lt2), e2 = __builtin_elementwise_fma(ng2, st.bb, lb2);
            score(SE + 1 + 2 * p, exp2f_(e1.x) + exp2f_(e2.x));
            score(SE + 2 + 2 * p, exp2f_(e1.y) + exp2f_(e2.y));
        };
        constexpr int NP = NA / 2;
        constexpr int D = (QB_GT_DEPTH + 1) / 2 < NP ? (QB_GT_DEPTH + 1) / 2 : NP;
        Stage ring[D + 1];
#pragma unroll
        for (int e = 0; e < D; ++e) ring[e] = issue(e);
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            if (e + D < NP) ring[(e + D) % (D + 1)] = issue(e + D);
            __builtin_amdgcn_sched_barrier(0);
            finish(e, ring[e % (D + 1)]);
            __builtin_amdgcn_sched_barrier(0);
        }
        return acc;
    } else {
        struct Stage {
            float4 kk;
            float bb;
        };
        auto issue = [&](int t) -> Stage {
            const int j = t > SE ? t - SE : SE - t;
            Stage st;
            st.kk = row[(j - 1) * NSEG];
            st.bb = L->blood_B[t];
            return st;
        };
        auto finish = [&](int t, const Stage& st) {
            const float F = fmaf(fmaf(fmaf(st.kk.w, f, st.kk.z), f, st.kk.y), f, st.kk.x);
            score(t, exp2f_(fmaf(fv.nd, F, lt)) + exp2f_(fmaf(fv.ng, st.bb, lb)));
        };
        auto tau_of = [](int e) { return e < NA ? SE + 1 + e : e - NA; };
        constexpr int kDepth = T <= 16 ? QB_GT_DEPTH : QB_GT_DEPTH_LONG;
        constexpr int D = kDepth < NE ? kDepth : NE;
        if (qb_phase_fence()) {   // a basic block of its own: see the x-indexed form below
        Stage ring[D + 1];
#pragma unroll
        for (int e = 0; e < D; ++e) ring[e] = issue(tau_of(e));
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e + D < NE) ring[(e + D) % (D + 1)] = issue(tau_of(e + D));
            __builtin_amdgcn_sched_barrier(0);
            finish(tau_of(e), ring[e % (D + 1)]);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        return acc;
    }
}

template <int T, int SE, bool MIR = false>
__device__ __forceinline__ float sample_sq_fast(const FwdLds* L, const QbDev& c,
                                                const VoxelLik<T>& k, float oef, float dbv) {
    const FwdFast fv = fwd_fast(c, oef, dbv);
    float acc = 0.0f;
    if (SE >= 0) {
        // Spin-echo signal first, then each tau's residual as soon as its signal exists: no T-element
        // signal array is kept live.  The per-draw factors of the normalised prediction
        //   yhat_t = (tissue_w 2^(nd F_t) + blood_w 2^(ng B_t)) / (s_se + 1e-3)          model.py:545
        // go into the exponents (two log2 per draw instead of three multiplies per tau), and the data
        // arrive pre-divided by sigma (prepare_lik<.., PRESCALE>): r_t = yt_t / s_t - yhat_t / s_t.
        // MIR (tau = 0 at the spin echo and the grid mirrors about it: the reference's protocols): x = 0 there,
        // F(0) = 0 exactly (the table's first coefficient), so the tissue factor is tissue_w and no lookup is needed
        // -- bit-identical.
        constexpr int kSE = SE >= 0 ? SE : 0;
        const float s_se = MIR ? fmaf(fv.tissue_w, 1.0f, fv.blood_w * exp2f_(fv.ng * c.blood_B[kSE]))
                               : fwd_signal_fast(L, c, fv, kSE);
        const float inv_np = rcpf_(s_se + 1e-3f);
        const float lt = log2f_(fv.tissue_w * inv_np), lb = log2f_(fv.blood_w * inv_np);  // log2(0) = -inf: term vanishes
        auto residual = [&](int t, float yh) {
            const float r = fmaf(-yh, k.inv_s[t], k.yt[t]);
            acc = fmaf(r, r, acc);
        };
        residual(kSE, s_se * inv_np);
        // The evaluated taus: MIR -- SE + 1 .. T - 1 (a mirrored pair is evaluated once, at its tau above the spin
        // echo, and scored against the pair's merged data point, prepare_lik<.., MIR>; 17 evaluations instead of 24
        // for tau = -28 .. 64 ms), then the taus below the spin echo that have no partner on the grid; otherwise every
        // tau but SE.  A ring keeps QB_X_DEPTH table rows in flight, and sched_barrier pins requests AND arithmetic:
        // left to itself the compiler hoists all T row reads to the top of the draw and spills (with straight-line
        // code and only a memory clobber every two taus the 24-tau kernel took 550 bytes of scratch per lane).
        constexpr int NA = T - 1 - kSE, NB = (2 * kSE - (T - 1)) > 0 ? 2 * kSE - (T - 1) : 0;
        constexpr int NE = MIR ? NA + NB : T - 1;
        auto tau_of = [](int e) { return MIR ? (e < NA ? kSE + 1 + e : e - NA) : (e < kSE ? e : e + 1); };
        struct Stage {
            float4 kk;
            float f, bb;
        };
        auto issue = [&](int t) -> Stage {
            Stage st;
            const float u = fabsf(fmaf((float)t, fv.ub, fv.ua));
            st.kk = L->tab[(int)u];
            st.f = __builtin_amdgcn_fractf(u);
            st.bb = L->blood_B[t];
            return st;
        };
        auto finish = [&](int t, const Stage& st) {
            const float F = fmaf(fmaf(fmaf(st.kk.w, st.f, st.kk.z), st.f, st.kk.y), st.f, st.kk.x);
            residual(t, exp2f_(fmaf(fv.nd, F, lt)) + exp2f_(fmaf(fv.ng, st.bb, lb)));
        };
        constexpr int D = QB_X_DEPTH < NE ? QB_X_DEPTH : NE;
        // (behind an opaque always-taken branch: as a basic block of its own the tau loop is scheduled and
        // register-allocated apart from the draw's prologue -- 472 -> 164 bytes of scratch at T = 24, 36 -> 0 at T = 11)
        if (qb_phase_fence()) {
        Stage ring[D + 1];
#pragma unroll
        for (int e = 0; e < D; ++e) ring[e] = issue(tau_of(e));
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e + D < NE) ring[(e + D) % (D + 1)] = issue(tau_of(e + D));
            __builtin_amdgcn_sched_barrier(0);
            finish(tau_of(e), ring[e % (D + 1)]);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        return acc;
    }
    float s[T];
#pragma unroll
    for (int t = 0; t < T; ++t) s[t] = fwd_signal_fast(L, c, fv, t);
    const float inv_np = rcpf_(se_norm<T, SE>(c, s));
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float r = fmaf(-s[t], inv_np, k.yt[t]) * k.inv_s[t];
        acc = fmaf(r, r, acc);
    }
    return acc;
}

// swr_p - swr_q of one draw: the only draw-dependent part of log q - log p (the Jacobian and the
// log-determinants cancel or are constant per voxel).
//
// The reference forms the logits by a round trip through the unit interval
// (model.py:393-396): x = clip((y - min)/range, 1e-6, 1 - 1e-6), logit(x) = log(x / (1 - x)),
// with y = sigmoid(a) * range + min.  For an unclipped draw that is the identity, logit(x) = a up
// to float32 rounding of the round trip; the clip of x at [1e-6, 1 - 1e-6] is a clip of the logit
// at +-log((1 - 1e-6) / 1e-6) = +-13.815510.  The fast path therefore clips the logit directly
// (no sigmoid, log or reciprocal per draw); the literal / generic path keeps the round trip.
#define QB_LOGIT_CLIP 13.815509557963774f
__device__ __forceinline__ float kl_swr_diff(const LogitMvn& q, const LogitMvn& p, float z0,
                                             float z1) {
    float a, b;
    reparam_logits(q, z0, z1, a, b);
    const float l0 = clampf_(a, -QB_LOGIT_CLIP, QB_LOGIT_CLIP);
    const float l1 = clampf_(b, -QB_LOGIT_CLIP, QB_LOGIT_CLIP);
    const float rq0 = l0 - q.mu_o, rq1 = l1 - q.mu_d;
    const float rp0 = l0 - p.mu_o, rp1 = l1 - p.mu_d;
    const float wq0 = rq0 * q.i_so, wq1 = fmaf(rq1, q.i_sd, rq0 * q.i_bl);
    const float wp0 = rp0 * p.i_so, wp1 = fmaf(rp1, p.i_sd, rp0 * p.i_bl);
    return fmaf(wp0, wp0, wp1 * wp1) - fmaf(wq0, wq0, wq1 * wq1);
}

// The K KL draws of one voxel restricted to this lane's share, fast path: returns sum over the draws of
// (swr_p - swr_q); n_kl = draws taken.  zk: explicit normals [K][2] or nullptr for the Philox stream.
//
// Whitened form (in-kernel Philox normals).  A draw y = mu_q + L_q z has, under q itself, the whitened residual z
// exactly, and under the prior  L_p^-1 (mu_q - mu_p) + (L_p^-1 L_q) z = d + M z  with d and the lower-triangular M
// fixed per voxel: swr_p - swr_q = |d + M z|^2 - |z|^2, seven instructions per draw instead of nineteen, no logits
// formed (and none of their cancellation: against the float64 oracle this form is at 2e-6 where the general one --
// the reference's arithmetic -- is at 8e-4 for far-apart q and prior).  Valid while the clip of the logits at
// +-13.8155 (model.py:393-396) cannot bind: Box-Muller on u1 >= 2^-17 bounds |z| by 4.8549 (QB_Z_MAX), so it cannot
// when
// |mu| + 4.8549 (|c| + e^s) stays below the clip for both logits (rounds 1-3: 6.7636 with 32-bit uniforms).  Decided per
// wave (any lane over the bound, or explicit normals: the general loop for all).
__device__ __forceinline__ float kl_draws_fast(const LogitMvn& q, const LogitMvn& prior, int K,
                                               const float* __restrict__ zk, uint64_t seed, uint64_t vox, int part,
                                               int& n_kl) {
    constexpr float kZMax = QB_Z_MAX;
    const float reach = fmaxf(fabsf(q.mu_o) + kZMax * q.e_so, fabsf(q.mu_d) + kZMax * (fabsf(q.c) + q.e_sd));
    float kl_sum = 0.0f;
    if (zk == nullptr && __all(reach < QB_LOGIT_CLIP)) {
        const float dmu_o = q.mu_o - prior.mu_o, dmu_d = q.mu_d - prior.mu_d;
        const float d0 = dmu_o * prior.i_so, m00 = q.e_so * prior.i_so;
        const float d1 = fmaf(dmu_d, prior.i_sd, dmu_o * prior.i_bl);
        const float m10 = fmaf(q.c, prior.i_sd, q.e_so * prior.i_bl), m11 = q.e_sd * prior.i_sd;
        // |d + M z|^2 - |z|^2 is quadratic in z with coefficients fixed per voxel, so the loop only gathers the draws'
        // five moments (sum z0, z1, z0^2, z1^2, z0 z1: five instructions per draw beside its normals) and the sum is
        // assembled once; a draw beyond K comes out of normals8 as z = 0 and adds nothing to any moment.
        float s0 = 0.0f, s1 = 0.0f, s00 = 0.0f, s11 = 0.0f, s01 = 0.0f;
        for (int g = part; 4 * g < K; g += QB_LANES_PER_VOXEL) {
            float z[8];
            const int cnt = K - 4 * g < 4 ? K - 4 * g : 4;
            n_kl += cnt;
            normals8_unscaled(seed, vox, (uint32_t)g, STREAM_KL, cnt, z);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                s0 += z[2 * d];
                s1 += z[2 * d + 1];
                s00 = fmaf(z[2 * d], z[2 * d], s00);
                s11 = fmaf(z[2 * d + 1], z[2 * d + 1], s11);
                s01 = fmaf(z[2 * d], z[2 * d + 1], s01);
            }
        }
        // the normals' common factor sqrt(2 ln 2) (normals8_unscaled), applied to the moments
        s0 *= QB_BM_K;
        s1 *= QB_BM_K;
        s00 *= QB_BM_K * QB_BM_K;
        s11 *= QB_BM_K * QB_BM_K;
        s01 *= QB_BM_K * QB_BM_K;
        // sum_k (d0 + m00 z0)^2 + (d1 + m10 z0 + m11 z1)^2 - z0^2 - z1^2
        const float nk = (float)n_kl;
        const float quad = fmaf(fmaf(m00, m00, m10 * m10) - 1.0f, s00, fmaf(fmaf(m11, m11, -1.0f), s11, 2.0f * m10 * m11 * s01));
        const float lin = 2.0f * fmaf(fmaf(d0, m00, d1 * m10), s0, d1 * m11 * s1);
        return fmaf(nk, fmaf(d0, d0, d1 * d1), lin + quad);
    }
    for (int g = part; 4 * g < K; g += QB_LANES_PER_VOXEL) {
        const int cnt = K - 4 * g < 4 ? K - 4 * g : 4;
        n_kl += cnt;
        DrawQuad dq;
        if (!zk) dq.load(seed, vox, (uint32_t)g, STREAM_KL);
#pragma unroll 1
        for (int d = 0; d < cnt; ++d) {
            float z0, z1;
            if (zk) {
                z0 = zk[2 * (4 * g + d)];
                z1 = zk[2 * (4 * g + d) + 1];
            } else {
                dq.next(z0, z1);
            }
            kl_sum += kl_swr_diff(q, prior, z0, z1);
        }
    }
    return kl_sum;
}

// The two Monte-Carlo sums of one voxel restricted to this lane's share of the draws.
//   nll_sum = sum over this half's likelihood draws of the per-draw NLL
//   kl_sum  = sum over this half's KL draws of log q(y) - log p(y)          model.py:596-603
// zs / zk: explicit normals of this voxel ([S][2] / [K][2]) or nullptr for the Philox stream.
// FAST: requires c.full_model, table mode, !predict_log, !use_student_t (checked on the host).
// MIR: `lik` was prepared with merged mirror pairs (prepare_lik<.., MIR>; always with the per-tau table).
template <int T, int SE, bool FAST, bool LITERAL, bool MIR = false, class LDS>
__device__ __forceinline__ void voxel_mc_sums(const LDS* L, const QbDev& c,
                                              const VoxelLik<T>& lik, const LogitMvn& q,
                                              const float* __restrict__ prior_row, int S, int K,
                                              const float* __restrict__ zs,
                                              const float* __restrict__ zk, uint64_t seed,
                                              uint64_t vox, int part, float& nll_sum,
                                              float& kl_sum) {
    nll_sum = 0.0f;
    kl_sum = 0.0f;
    int n_lik = 0, n_kl = 0;  // draws taken by this lane
    // Across the likelihood loop only what the reparameterisation reads stays live (mu, e^s, c) plus the one number
    // the KL's constant needs; the inverse scales come back from two reciprocals afterwards (five registers less
    // through the loop that owns the register budget).
    const float q_s_sum = q.s_o + q.s_d;
    __builtin_amdgcn_s_setprio(QB_PRIO_LIK);
    // This lane's draws: Philox calls part, part + 4, ... -> draws 4 g .. 4 g + 3 each, the last call possibly short.
    // ONE loop over them (a call's words are refilled every fourth trip): nested as calls x draws the register
    // allocator split far more live ranges around the inner loop (612 against 88 bytes of scratch at T = 24).
    {
        const int calls = S > 4 * part ? (S - 4 * part + 15) / 16 : 0;           // calls g = part + 4 k < ceil(S / 4)
        const int last = calls > 0 ? S - 4 * (part + 4 * (calls - 1)) : 0;      // draws of the last call: 1 .. 4
        n_lik = calls > 0 ? 4 * (calls - 1) + (last < 4 ? last : 4) : 0;
        DrawQuad dq;
        uint32_t g = (uint32_t)part;
#pragma unroll 1
        for (int i = 0; i < n_lik; ++i) {
            float z0, z1;
            if (zs) {
                const int draw = 4 * (part + 4 * (i >> 2)) + (i & 3);
                z0 = zs[2 * draw];
                z1 = zs[2 * draw + 1];
            } else {
                if ((i & 3) == 0) {
                    dq.load(seed, vox, g, STREAM_LIK);
                    g += QB_LANES_PER_VOXEL;
                }
                dq.next(z0, z1);
            }
            float a, b, oef = 0.0f, dbv = 0.0f;
            reparam_logits(q, z0, z1, a, b);
            if constexpr (!(FAST && IsGtLds<LDS>::value)) forward_transform(a, b, oef, dbv);
            if constexpr (FAST && IsGtLds<LDS>::value) {
                static_assert(!IsGtLds<LDS>::value || MIR, "the per-tau table scores merged mirror pairs");
                nll_sum += sample_sq_fast<T, SE>(L, c, lik, sigmoidf_(a), sigmoidf_(b));
            } else if constexpr (FAST) {
                nll_sum += sample_sq_fast<T, SE, MIR>(L, c, lik, oef, dbv);
            } else {
                nll_sum += sample_nll<T, SE, LITERAL>(L, c, lik, oef, dbv);
            }
        }
    }
    if (FAST) {  // sum_d [0.5 sum_t r^2 + sum_t log sigma + T log sqrt(2 pi)] over this lane's draws
        nll_sum = fmaf(0.5f, nll_sum, (float)n_lik * lik.log_s_sum);
    }
    // the prior's parameters are fetched and transformed only now: they are dead weight in
    // registers during the likelihood loop
    float pv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) pv[k] = prior_row[k];
    const LogitMvn prior = make_mvn(pv);
    __builtin_amdgcn_s_setprio(QB_PRIO_KL);
    if constexpr (FAST) {
        LogitMvn qk;
        qk.mu_o = q.mu_o; qk.mu_d = q.mu_d; qk.c = q.c; qk.e_so = q.e_so; qk.e_sd = q.e_sd;
        qk.s_o = q_s_sum; qk.s_d = 0.0f;                  // only their sum is used below
        qk.i_so = rcpf_(q.e_so);
        qk.i_sd = rcpf_(q.e_sd);
        qk.i_bl = -(qk.i_so * qk.i_sd) * q.c;
        kl_sum = kl_draws_fast(qk, prior, K, zk, seed, vox, part, n_kl);
    } else {
        for (int j = part; 2 * j < K; j += QB_LANES_PER_VOXEL) {
            float z[4];
            const bool two = 2 * j + 1 < K;
            n_kl += two ? 2 : 1;
            if (zk) {
                z[0] = zk[4 * j];
                z[1] = zk[4 * j + 1];
                z[2] = two ? zk[4 * j + 2] : 0.0f;
                z[3] = two ? zk[4 * j + 3] : 0.0f;
            } else {
                normals4(seed, vox, (uint32_t)j, STREAM_KL, z);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                if (d == 1 && !two) break;
                float a, b, oef, dbv;
                reparam_logits(q, z[2 * d], z[2 * d + 1], a, b);  // create_samples, model.py:318-324
                forward_transform(a, b, oef, dbv);
                const LogitObs o = make_obs(oef, dbv);
                // log_q - log_p with log = -nlogp                 model.py:596-597, 603
                kl_sum += nlogp(o, prior) - nlogp(o, q);
            }
        }
    }
    __builtin_amdgcn_s_setprio(QB_PRIO_AFTER);
    if (FAST) {  // log q - log p = 0.5 (swr_p - swr_q) + (s_o + s_d)_p - (s_o + s_d)_q per draw
        kl_sum = fmaf(0.5f, kl_sum, (float)n_kl * ((prior.s_o + prior.s_d) - q_s_sum));
    }
}

// Block-level reduction of the three masked sums into one double3 partial per workgroup.
// red: LDS scratch of 3 * (blockDim.x / 64) doubles.
__device__ __forceinline__ void block_partials(double* red, float s_nll, float s_kl, float s_m,
                                               double* __restrict__ partials) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    s_nll = wave_sum(s_nll);
    s_kl = wave_sum(s_kl);
    s_m = wave_sum(s_m);
    if (lane == 0) {
        red[3 * wave + 0] = (double)s_nll;
        red[3 * wave + 1] = (double)s_kl;
        red[3 * wave + 2] = (double)s_m;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a = 0.0;
        for (int w = 0; w < nw; ++w) a += red[3 * w + threadIdx.x];
        partials[3 * blockIdx.x + threadIdx.x] = a;
    }
}

// Final deterministic pass: sums[k] = sum_b partials[b][k].  One block of 192 threads.
static __global__ void reduce_partials_kernel(const double* __restrict__ partials, int nblocks,
                                       double* __restrict__ sums) {
    __shared__ double sh[192];
    const int k = threadIdx.x / 64, lane = threadIdx.x & 63;
    double a = 0.0;
    for (int b = lane; b < nblocks; b += 64) a += partials[3 * b + k];
    sh[threadIdx.x] = a;
    __syncthreads();
    if (lane == 0) {
        double t = 0.0;
        for (int i = 0; i < 64; ++i) t += sh[64 * k + i];
        sums[k] = t;
    }
}

}  // namespace qb

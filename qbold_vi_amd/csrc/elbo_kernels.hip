// elbo_kernels.hip -- stand-alone ELBO kernel (given encoder outputs) and the small logit-Normal
// utilities of the reference's API surface (reparameterised sampling, log-density, posterior
// moments).  The per-voxel arithmetic is in elbo_core.h / qbold_dev.h.
//
// Roofline (qbold_elbo_fwd, T=11): reads 4T+4+20+20+4T = 132 B per voxel, writes 8 B; work is
// S forward-model evaluations + K log-density pairs + (S+K)/2 Philox calls per voxel
// (~25 kFLOP-equivalent at S=32, K=70) -> f32 VALU-bound by two orders of magnitude.
#include <cmath>

#include "elbo_core.h"
#include "qbold_ctx.h"

namespace {

constexpr int kBlock = 256;              // 4 waves
constexpr int kVoxPerBlock = kBlock / QB_LANES_PER_VOXEL;  // 16 voxels per wave, 4 lanes each

template <int T, int SE, bool GT>
struct ElboLds { using type = qb::FwdLds; };
template <int T, int SE>
struct ElboLds<T, SE, true> { using type = qb::GtLds<T, SE>; };

// GT: per-tau OEF-indexed table (GtLds) instead of the x-indexed one; needs FAST and a compile-time spin echo.
// MIR: the protocol mirrors about the spin echo (qbold_ctx::grid_mirrors): mirrored tau pairs are evaluated once and
// scored as one merged data point (elbo_core.h, prepare_lik); GT implies it.
template <int T, int SE, bool FAST, bool LITERAL, bool GT = false, bool MIR = false>
__global__ __launch_bounds__(kBlock) void elbo_fwd_kernel(
    QbDev c, const float4* __restrict__ g_tab, const float* __restrict__ x,
    const float* __restrict__ mask, const float* __restrict__ q, const float* __restrict__ prior,
    const float* __restrict__ sigma, const float* __restrict__ zs, const float* __restrict__ zk,
    int S, int K, uint64_t seed, int64_t voxel0, float2* __restrict__ nll_kl,
    double* __restrict__ partials, int64_t N) {
    // same choice of sampling table as the fused kernel (vi_kernels.hip), so that the fused and the unfused
    // evaluation of a voxel run the same arithmetic
    static_assert(!GT || (FAST && SE >= 0 && qb::gtab_segs(T) > 0), "GT needs the fast path with a compile-time spin echo");
    static_assert(!MIR || (FAST && SE >= 0), "merged mirror pairs: fast path with a compile-time spin echo");
    constexpr bool kMir = GT || MIR;
    using Lds = typename ElboLds<T, SE, GT>::type;
    __shared__ Lds L;
    __shared__ double red[3 * (kBlock / 64)];
    if constexpr (qb::IsGtLds<Lds>::value) {
        qb::gt_lds_fill(&L, g_tab, c);
    } else {
        qb::fwd_lds_fill(&L, g_tab, true);
        if (threadIdx.x < QB_MAX_T) L.blood_B[threadIdx.x] = c.blood_B[threadIdx.x];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    float s_nll = 0.0f, s_kl = 0.0f, s_m = 0.0f;
    const int64_t ntile = (N + kVoxPerBlock - 1) / kVoxPerBlock;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t v = tile * kVoxPerBlock + wave * QB_VOX_PER_WAVE + (lane & 15);
        if (v < N) {
            float xv[T], sv[T], qv[5];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                xv[t] = x[v * T + t];
                sv[t] = sigma[v * T + t];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) qv[i] = q[v * 5 + i];
            const float m = mask ? mask[v] : 1.0f;
            qb::VoxelLik<T> lik;
            qb::prepare_lik<T, SE, false, (FAST && SE >= 0), FAST, kMir>(c, xv, sv, m, lik);
            const qb::LogitMvn qm = qb::make_mvn(qv);
            float nll_part, kl_part;
            qb::voxel_mc_sums<T, SE, FAST, LITERAL, kMir>(&L, c, lik, qm, prior + v * 5, S, K, zs ? zs + v * S * 2 : nullptr,
                                          zk ? zk + v * K * 2 : nullptr, seed,
                                          (uint64_t)(voxel0 + v), part, nll_part, kl_part);
            // the four lanes of a voxel are active together (v depends on lane & 15 only)
            const float nll = qb::voxel_sum(nll_part) / (float)S;
            const float kl = K > 0 ? qb::voxel_sum(kl_part) / (float)K : 0.0f;
            if (part == 0) {
                if (nll_kl) nll_kl[v] = make_float2(nll, kl);
                s_nll += nll * m;               // model.py:564
                s_kl += m > 0.0f ? kl : 0.0f;   // model.py:661
                s_m += m;
            }
        }
    }
    qb::block_partials(red, s_nll, s_kl, s_m, partials);
}

// Any number of taus (T <= 64; BASELINE config 3 uses 64): the per-voxel normalised data and inverse
// sigmas live in LDS ([t][voxel], conflict-free) instead of registers, the tau loop is a run-time
// loop.  Same lane mapping, Philox stream and arithmetic as the fast register path.
constexpr int kGenBlock = 128;
constexpr int kGenVox = kGenBlock / QB_LANES_PER_VOXEL;

__global__ __launch_bounds__(kGenBlock) void elbo_fwd_generic_kernel(
    QbDev c, const float4* __restrict__ g_tab, const float* __restrict__ x,
    const float* __restrict__ mask, const float* __restrict__ q, const float* __restrict__ prior,
    const float* __restrict__ sigma, const float* __restrict__ zs, const float* __restrict__ zk,
    int S, int K, uint64_t seed, int64_t voxel0, float2* __restrict__ nll_kl,
    double* __restrict__ partials, int64_t N) {
    extern __shared__ __align__(16) unsigned char smem[];
    qb::FwdLds* L = reinterpret_cast<qb::FwdLds*>(smem);
    float* yt = reinterpret_cast<float*>(smem + sizeof(qb::FwdLds));  // [T][kGenVox]
    float* is = yt + QB_MAX_T * kGenVox;                               // [T][kGenVox]
    __shared__ double red[3 * (kGenBlock / 64)];
    qb::fwd_lds_fill(L, g_tab, false);
    if (threadIdx.x < QB_MAX_T) L->blood_B[threadIdx.x] = c.blood_B[threadIdx.x];
    __syncthreads();

    const int T = c.T, se = c.se_idx;
    // tau = 0 at the spin echo and one-image normalisation (the reference's protocols): the signal is even
    // in tau, so each pair tau_{se-j} = -tau_{se+j} is evaluated once, and the per-draw factors go into the
    // exponents as in sample_sq_fast
    const bool mirrored = !c.multi_norm && fmaf((float)se, c.tauh_step, c.tauh0) == 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    const int vl = wave * QB_VOX_PER_WAVE + (lane & 15);  // voxel slot inside the block
    float s_nll = 0.0f, s_kl = 0.0f, s_m = 0.0f;
    const int64_t ntile = (N + kGenVox - 1) / kGenVox;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t v = tile * kGenVox + vl;
        const bool live = v < N;
        const int64_t vc = live ? v : N - 1;
        const float* xv = x + vc * T;
        const float* sv = sigma + vc * T;
        const float nt = c.multi_norm ? (xv[se - 1] + xv[se] + xv[se + 1]) / 3.0f + 1e-3f : xv[se] + 1e-3f;
        const float inv_nt = qb::rcpf_(nt);
        float ls = 0.0f;
        __syncthreads();  // previous tile's readers are done
        for (int t = part; t < T; t += QB_LANES_PER_VOXEL) {
            yt[t * kGenVox + vl] = xv[t] * inv_nt;
            is[t * kGenVox + vl] = qb::rcpf_(sv[t]);
            ls += QB_LN2 * qb::log2f_(sv[t]);
        }
        const float log_s_sum = qb::voxel_sum(ls) + (float)T * 0.9189385332046727f;
        __syncthreads();
        if (live) {
            float qv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) qv[i] = q[v * 5 + i];
            const float m = mask ? mask[v] : 1.0f;
            const qb::LogitMvn qm = qb::make_mvn(qv);
            const uint64_t vox = (uint64_t)(voxel0 + v);
            const float* zsv = zs ? zs + v * S * 2 : nullptr;
            const float* zkv = zk ? zk + v * K * 2 : nullptr;
            float nll_sum = 0.0f, kl_sum = 0.0f;
            int n_lik = 0, n_kl = 0;
            for (int g = part; 4 * g < S; g += QB_LANES_PER_VOXEL) {
                const int cnt = S - 4 * g < 4 ? S - 4 * g : 4;
                n_lik += cnt;
                qb::DrawQuad dq;
                if (!zsv) dq.load(seed, vox, (uint32_t)g, qb::STREAM_LIK);
#pragma unroll 1
                for (int d = 0; d < cnt; ++d) {
                    float z0, z1;
                    if (zsv) {
                        z0 = zsv[2 * (4 * g + d)];
                        z1 = zsv[2 * (4 * g + d) + 1];
                    } else {
                        dq.next(z0, z1);
                    }
                    float a, b, oef, dbv;
                    qb::reparam_logits(qm, z0, z1, a, b);
                    qb::forward_transform(a, b, oef, dbv);
                    const qb::FwdFast fv = qb::fwd_fast(c, oef, dbv);
                    if (mirrored) {
                        const float s_se = fmaf(fv.tissue_w, 1.0f, fv.blood_w * qb::exp2f_(fv.ng * L->blood_B[se]));
                        const float inv_np = qb::rcpf_(s_se + 1e-3f);
                        const float lt = qb::log2f_(fv.tissue_w * inv_np), lb = qb::log2f_(fv.blood_w * inv_np);
                        auto signal = [&](int t) {
                            const float u = fabsf(fmaf((float)t, fv.ub, fv.ua));
                            const float4 kk = L->tab[(int)u];
                            const float f = __builtin_amdgcn_fractf(u);
                            const float F = fmaf(fmaf(fmaf(kk.w, f, kk.z), f, kk.y), f, kk.x);
                            return qb::exp2f_(fmaf(fv.nd, F, lt)) + qb::exp2f_(fmaf(fv.ng, L->blood_B[t], lb));
                        };
                        float acc = 0.0f;
                        auto residual = [&](int t, float yh) {
                            const float r = (yt[t * kGenVox + vl] - yh) * is[t * kGenVox + vl];
                            acc = fmaf(r, r, acc);
                        };
                        residual(se, s_se * inv_np);
                        for (int t = se + 1; t < T; ++t) {
                            const float yh = signal(t);
                            residual(t, yh);
                            if (2 * se - t >= 0) residual(2 * se - t, yh);
                        }
                        for (int t = 0; t < 2 * se - (T - 1); ++t) residual(t, signal(t));  // no partner on the grid
                        nll_sum += acc;
                        continue;
                    }
                    float np_ = qb::fwd_signal_fast(L, c, fv, se);
                    if (c.multi_norm)
                        np_ = (np_ + qb::fwd_signal_fast(L, c, fv, se - 1) + qb::fwd_signal_fast(L, c, fv, se + 1)) / 3.0f;
                    const float inv_np = qb::rcpf_(np_ + 1e-3f);
                    float acc = 0.0f;
                    for (int t = 0; t < T; ++t) {
                        const float st = qb::fwd_signal_fast(L, c, fv, t);
                        const float r = fmaf(-st, inv_np, yt[t * kGenVox + vl]) * is[t * kGenVox + vl];
                        acc = fmaf(r, r, acc);
                    }
                    nll_sum += acc;
                }
            }
            nll_sum = fmaf(0.5f, nll_sum, (float)n_lik * log_s_sum);
            float pv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) pv[i] = prior[v * 5 + i];
            const qb::LogitMvn pm = qb::make_mvn(pv);
            kl_sum = qb::kl_draws_fast(qm, pm, K, zkv, seed, vox, part, n_kl);
            kl_sum = fmaf(0.5f, kl_sum, (float)n_kl * ((pm.s_o + pm.s_d) - (qm.s_o + qm.s_d)));
            const float nll = qb::voxel_sum(nll_sum) / (float)S;
            const float kl = K > 0 ? qb::voxel_sum(kl_sum) / (float)K : 0.0f;
            if (part == 0) {
                if (nll_kl) nll_kl[v] = make_float2(nll, kl);
                s_nll += nll * m;
                s_kl += m > 0.0f ? kl : 0.0f;
                s_m += m;
            }
        }
    }
    qb::block_partials(red, s_nll, s_kl, s_m, partials);
}

// Long protocols with a compile-time tau count and spin-echo index (BASELINE config 3: T = 64, tau = 0 at
// index 12).  2 T values per voxel do not fit the register budget of a VALU-bound kernel (as registers the
// T = 64 instance of elbo_fwd_kernel takes 257 VGPRs: one wave per SIMD), so the pre-scaled data and the
// inverse sigmas live in LDS as float2 rows [t][voxel] -- one conflict-free ds_read_b64 per scored tau at an
// immediate offset -- and everything else is the fast path of elbo_core.h: the tau loop unrolled, mirrored
// pairs evaluated once (52 signal evaluations per draw for 64 taus), per-draw factors in the exponents.
// A voxel's row is read as four float4 per lane (the four lanes of a voxel own a quarter of the taus each).
// LOGSIG: `sigma` holds log sigma, the sigma head before its exp (model.py:211-214).
#ifndef QB_LDS_PIPE
#define QB_LDS_PIPE 1
#endif
#ifndef QB_LDS_DEPTH
#define QB_LDS_DEPTH 2
#endif
constexpr int kLdsBlock = 256;
constexpr int kLdsVox = kLdsBlock / QB_LANES_PER_VOXEL;

template <int T, int SE, bool LOGSIG>
__global__ __launch_bounds__(kLdsBlock) void elbo_fwd_lds_kernel(
    QbDev c, const float4* __restrict__ g_tab, const float* __restrict__ x,
    const float* __restrict__ mask, const float* __restrict__ q, const float* __restrict__ prior,
    const float* __restrict__ sigma, const float* __restrict__ zs, const float* __restrict__ zk,
    int S, int K, uint64_t seed, int64_t voxel0, float2* __restrict__ nll_kl,
    double* __restrict__ partials, int64_t N) {
    static_assert(T % 16 == 0 && SE >= 0 && SE < T, "a lane owns T / 4 taus as float4 loads");
    __shared__ qb::FwdLds L;
    __shared__ float2 dat[T * kLdsVox];  // (y_t / sigma_t, 1 / sigma_t) at [t][voxel slot]
    __shared__ double red[3 * (kLdsBlock / 64)];
    qb::fwd_lds_fill(&L, g_tab, false);
    if (threadIdx.x < QB_MAX_T) L.blood_B[threadIdx.x] = c.blood_B[threadIdx.x];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    const int vl = wave * QB_VOX_PER_WAVE + (lane & 15);
    float2* my = dat + vl;
    float s_nll = 0.0f, s_kl = 0.0f, s_m = 0.0f;
    const int64_t ntile = (N + kLdsVox - 1) / kLdsVox;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t v = tile * kLdsVox + vl;
        const bool live = v < N;
        const int64_t vc = live ? v : N - 1;
        const float4* xq = reinterpret_cast<const float4*>(x + vc * T) + (T / 16) * part;
        const float4* sq = reinterpret_cast<const float4*>(sigma + vc * T) + (T / 16) * part;
        const float inv_nt = qb::rcpf_(x[vc * T + SE] + 1e-3f);  // model.py:545
        float ls = 0.0f;
        __syncthreads();  // the previous tile's draws are done with dat (first trip: the table is filled)
#pragma unroll
        for (int k4 = 0; k4 < T / 16; ++k4) {
            const float4 xx = xq[k4], ss = sq[k4];
            const float xa[4] = {xx.x, xx.y, xx.z, xx.w}, sa[4] = {ss.x, ss.y, ss.z, ss.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float is;
                if (LOGSIG) {
                    is = qb::exp2f_(-QB_LOG2E * sa[e]);
                    ls += sa[e];
                } else {
                    is = qb::rcpf_(sa[e]);
                    ls += QB_LN2 * qb::log2f_(sa[e]);
                }
                my[((T / 4) * part + 4 * k4 + e) * kLdsVox] = make_float2(xa[e] * inv_nt * is, is);
            }
        }
        __syncthreads();
        // Mirrored pairs as one data point (elbo_core.h, prepare_lik<.., MIR>): the signal is even in tau, so tau_{SE+j}
        // and tau_{SE-j} share one prediction; their two residuals are (Q - yh P)^2 + D with the merged point (Q, P)
        // stored at SE + j and D, which no draw changes, added to the per-draw constant.  A voxel's four lanes take
        // the pairs j = part + 1, part + 5, ... ; one ds_read_b64 and two FMAs less per pair and draw.
        float dsum = 0.0f;
#pragma unroll
        for (int j = 1; j <= SE && SE + j < T; ++j) {
            if (((j - 1) & 3) == part) {
                const float2 p1 = my[(SE + j) * kLdsVox], p2 = my[(SE - j) * kLdsVox];
                const float pp = fmaf(p1.y, p1.y, p2.y * p2.y);
                const float ip = __builtin_amdgcn_rsqf(pp);
                const float cr = fmaf(p1.x, p2.y, -(p2.x * p1.y)) * ip;
                my[(SE + j) * kLdsVox] = make_float2(fmaf(p1.x, p1.y, p2.x * p2.y) * ip, pp * ip);
                dsum = fmaf(cr, cr, dsum);
            }
        }
        const float log_s_sum = qb::voxel_sum(fmaf(0.5f, dsum, ls)) + (float)T * 0.9189385332046727f;  // + T log sqrt(2 pi)
        __syncthreads();
        if (live) {
            float qv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) qv[i] = q[v * 5 + i];
            const float m = mask ? mask[v] : 1.0f;
            const qb::LogitMvn qm = qb::make_mvn(qv);
            const uint64_t vox = (uint64_t)(voxel0 + v);
            const float* zsv = zs ? zs + v * S * 2 : nullptr;
            const float* zkv = zk ? zk + v * K * 2 : nullptr;
            float nll_sum = 0.0f, kl_sum = 0.0f;
            int n_lik = 0, n_kl = 0;
            __builtin_amdgcn_s_setprio(QB_PRIO_LIK);
            for (int g = part; 4 * g < S; g += QB_LANES_PER_VOXEL) {
                const int cnt = S - 4 * g < 4 ? S - 4 * g : 4;
                n_lik += cnt;
                qb::DrawQuad dq;
                if (!zsv) dq.load(seed, vox, (uint32_t)g, qb::STREAM_LIK);
#pragma unroll 1
                for (int d = 0; d < cnt; ++d) {
                    float z0, z1;
                    if (zsv) {
                        z0 = zsv[2 * (4 * g + d)];
                        z1 = zsv[2 * (4 * g + d) + 1];
                    } else {
                        dq.next(z0, z1);
                    }
                    float a, b, oef, dbv;
                    qb::reparam_logits(qm, z0, z1, a, b);
                    qb::forward_transform(a, b, oef, dbv);
                    const qb::FwdFast fv = qb::fwd_fast(c, oef, dbv);
                    // tau = 0 at the spin echo (checked by the host dispatch): F(0) = 0, no table row
                    const float s_se = fmaf(fv.tissue_w, 1.0f, fv.blood_w * qb::exp2f_(fv.ng * L.blood_B[SE]));
                    const float inv_np = qb::rcpf_(s_se + 1e-3f);
                    const float lt = qb::log2f_(fv.tissue_w * inv_np), lb = qb::log2f_(fv.blood_w * inv_np);
                    float acc = 0.0f;
#if QB_LDS_PIPE
                    // One-deep software pipeline over the taus: every LDS read an evaluation needs -- its table row (a
                    // random row per lane: the conflicted, slow read), the blood bracket and the voxel's data pair(s)
                    // -- is requested one evaluation ahead, so the round trip of the table read (~100+ cycles with its
                    // bank conflicts) overlaps the previous tau's polynomial, exponentials and residuals instead of
                    // being waited for right behind its issue (51 exposed round trips per draw before).
                    struct Stage {
                        float4 kk;
                        float f, bb;
                        float2 d0;
                    };
                    auto issue = [&](int t) -> Stage {
                        Stage st;
                        const float u = fabsf(fmaf((float)t, fv.ub, fv.ua));
                        st.kk = L.tab[(int)u];
                        st.f = __builtin_amdgcn_fractf(u);
                        st.bb = L.blood_B[t];
                        st.d0 = my[t * kLdsVox];     // t <= 2 SE: the merged pair (tau_t, tau_{2 SE - t})
                        return st;
                    };
                    auto finish = [&](int t, const Stage& st) {
                        const float F = fmaf(fmaf(fmaf(st.kk.w, st.f, st.kk.z), st.f, st.kk.y), st.f, st.kk.x);
                        const float yh = qb::exp2f_(fmaf(fv.nd, F, lt)) + qb::exp2f_(fmaf(fv.ng, st.bb, lb));
                        const float r = fmaf(-yh, st.d0.y, st.d0.x);
                        acc = fmaf(r, r, acc);
                    };
                    {
                        const float2 dd = my[SE * kLdsVox];
                        const float r = fmaf(-(s_se * inv_np), dd.y, dd.x);
                        acc = fmaf(r, r, acc);
                    }
                    static_assert(2 * SE - (T - 1) <= 0, "the pipelined form covers protocols whose taus below the "
                                                         "spin echo all have a mirror partner");
                    // QB_LDS_DEPTH evaluations ahead; sched_barrier pins BOTH the requests and the arithmetic (a memory
                    // clobber alone lets the compiler sink tau t's arithmetic behind the next requests and wait for a
                    // table row right behind its issue again)
                    constexpr int D = QB_LDS_DEPTH;
                    Stage ring[D + 1];
#pragma unroll
                    for (int k = 0; k < D; ++k) ring[k] = issue(SE + 1 + k);
#pragma unroll
                    for (int t = SE + 1; t < T; ++t) {
                        if (t + D < T) ring[(t - SE - 1 + D) % (D + 1)] = issue(t + D);
                        __builtin_amdgcn_sched_barrier(0);
                        finish(t, ring[(t - SE - 1) % (D + 1)]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#else
                    auto signal = [&](int t) -> float {
                        const float u = fabsf(fmaf((float)t, fv.ub, fv.ua));
                        const float4 kk = L.tab[(int)u];
                        const float f = __builtin_amdgcn_fractf(u);
                        const float F = fmaf(fmaf(fmaf(kk.w, f, kk.z), f, kk.y), f, kk.x);
                        return qb::exp2f_(fmaf(fv.nd, F, lt)) + qb::exp2f_(fmaf(fv.ng, L.blood_B[t], lb));
                    };
                    auto residual = [&](int t, float yh) {
                        const float2 dd = my[t * kLdsVox];
                        const float r = fmaf(-yh, dd.y, dd.x);
                        acc = fmaf(r, r, acc);
                    };
                    residual(SE, s_se * inv_np);
#pragma unroll
                    for (int t = SE + 1; t < T; ++t) {
                        residual(t, signal(t));      // t <= 2 SE: against the merged pair
                        if (((t - SE) % QB_LIK_BARRIER) == 0) asm volatile("" ::: "memory");
                    }
#pragma unroll
                    for (int t = 0; t < 2 * SE - (T - 1); ++t) residual(t, signal(t));  // no partner on the grid
#endif
                    nll_sum += acc;
                }
            }
            nll_sum = fmaf(0.5f, nll_sum, (float)n_lik * log_s_sum);
            float pv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) pv[i] = prior[v * 5 + i];
            const qb::LogitMvn pm = qb::make_mvn(pv);
            __builtin_amdgcn_s_setprio(QB_PRIO_KL);
            kl_sum = qb::kl_draws_fast(qm, pm, K, zkv, seed, vox, part, n_kl);
            __builtin_amdgcn_s_setprio(QB_PRIO_AFTER);
            kl_sum = fmaf(0.5f, kl_sum, (float)n_kl * ((pm.s_o + pm.s_d) - (qm.s_o + qm.s_d)));
            const float nll = qb::voxel_sum(nll_sum) / (float)S;
            const float kl = K > 0 ? qb::voxel_sum(kl_sum) / (float)K : 0.0f;
            if (part == 0) {
                if (nll_kl) nll_kl[v] = make_float2(nll, kl);
                s_nll += nll * m;
                s_kl += m > 0.0f ? kl : 0.0f;
                s_m += m;
            }
        }
    }
    qb::block_partials(red, s_nll, s_kl, s_m, partials);
}

// The same protocol on the per-tau OEF-indexed tissue table (qbold_dev.h, GtLds): the table coordinate -- segment index
// and fraction of NSEG sigmoid(a) -- is formed ONCE PER DRAW and serves all 51 evaluated taus, whose rows sit at immediate
// offsets from one address register: the coordinate FMA, v_cvt_i32, v_fract and address shift of the x-indexed table
// leave every (draw, tau) (4 of its 14 vector instructions).  The table (51 taus x 68 segments x 16 B = 54 KiB) is shared
// by ONE 1,024-thread workgroup per CU whose 256 voxels' merged data rows (52 x 256 x 8 B = 104 KiB; the mirrored pairs
// as one data point, elbo_core.h) fill the rest of the LDS.  A lane prepares the 16 taus it owns and merges its pairs in
// registers: pairs (13, 11) .. (15, 9) lie inside lane group 0's taus, pairs (16, 8) .. (24, 0) join group 1's taus with
// group 0's, which group 1 reads for itself (the same cache lines) -- no exchange, one barrier per tile and side.
constexpr int kGtBlock = 1024;
constexpr int kGtVox = kGtBlock / QB_LANES_PER_VOXEL;

template <int T, int SE, bool LOGSIG>
__global__ __launch_bounds__(kGtBlock) void elbo_fwd_gt64_kernel(
    QbDev c, const float4* __restrict__ g_gtab, const float* __restrict__ x,
    const float* __restrict__ mask, const float* __restrict__ q, const float* __restrict__ prior,
    const float* __restrict__ sigma, int S, int K, uint64_t seed, int64_t voxel0, float2* __restrict__ nll_kl,
    double* __restrict__ partials, int64_t N) {
    static_assert(T == 64 && SE == 12, "lane ownership of the taus below is written out for the 64-tau protocol");
    constexpr int NSEG = qb::gtab_segs(T), J = qb::gtab_taus(T, SE), NROW = T - SE;   // rows e = t - SE = 0 .. 51
    static_assert(J == NROW - 1 && NSEG > 0, "one table row set per evaluated tau above the spin echo");
    extern __shared__ __align__(16) unsigned char smem[];
    float4* gtab = reinterpret_cast<float4*>(smem);                                   // [J][NSEG]
    float* bloodB = reinterpret_cast<float*>(smem + sizeof(float4) * J * NSEG);       // [T]
    float2* dat = reinterpret_cast<float2*>(smem + sizeof(float4) * J * NSEG + sizeof(float) * T);   // [NROW][kGtVox]
    double* red = reinterpret_cast<double*>(smem + sizeof(float4) * J * NSEG + sizeof(float) * T +
                                            sizeof(float2) * NROW * kGtVox);
    for (int i = threadIdx.x; i < J * NSEG; i += kGtBlock) gtab[i] = g_gtab[i];
    if (threadIdx.x < T) bloodB[threadIdx.x] = c.blood_B[threadIdx.x];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    const int vl = wave * QB_VOX_PER_WAVE + (lane & 15);
    float2* my = dat + vl;
    float s_nll = 0.0f, s_kl = 0.0f, s_m = 0.0f;
    const int64_t ntile = (N + kGtVox - 1) / kGtVox;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t v = tile * kGtVox + vl;
        const bool live = v < N;
        const int64_t vc = live ? v : N - 1;
        const float inv_nt = qb::rcpf_(x[vc * T + SE] + 1e-3f);  // model.py:545
        // this lane's sixteen taus t = 16 part .. 16 part + 15 as (a, s) = (y / sigma, 1 / sigma)
        float a[16], sg[16], ls = 0.0f;
        {
            const float4* xq = reinterpret_cast<const float4*>(x + vc * T) + 4 * part;
            const float4* sq = reinterpret_cast<const float4*>(sigma + vc * T) + 4 * part;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float4 xx = xq[k4], ss = sq[k4];
                const float xa[4] = {xx.x, xx.y, xx.z, xx.w}, sa[4] = {ss.x, ss.y, ss.z, ss.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float is;
                    if (LOGSIG) {
                        is = qb::exp2f_(-QB_LOG2E * sa[e]);
                        ls += sa[e];
                    } else {
                        is = qb::rcpf_(sa[e]);
                        ls += QB_LN2 * qb::log2f_(sa[e]);
                    }
                    a[4 * k4 + e] = xa[e] * inv_nt * is;
                    sg[4 * k4 + e] = is;
                }
            }
        }
        // lane group 1 fetches taus 0 .. 8 (group 0's) once more for its pairs (16, 8) .. (24, 0)
        float am[9], sm[9];
        if (part == 1) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float sv = sigma[vc * T + t];
                const float is = LOGSIG ? qb::exp2f_(-QB_LOG2E * sv) : qb::rcpf_(sv);
                am[t] = x[vc * T + t] * inv_nt * is;
                sm[t] = is;
            }
        }
        __syncthreads();  // the previous tile's draws are done with dat (first trip: the table is filled)
        float dsum = 0.0f;
        auto merged = [&](float a1, float s1, float a2, float s2) {   // elbo_core.h, prepare_lik<.., MIR>
            const float pp = fmaf(s1, s1, s2 * s2);
            const float ip = __builtin_amdgcn_rsqf(pp);
            const float cr = fmaf(a1, s2, -(a2 * s1)) * ip;
            dsum = fmaf(cr, cr, dsum);
            return make_float2(fmaf(a1, s1, a2 * s2) * ip, pp * ip);
        };
        if (part == 0) {          // t = 12 (the spin echo) plain; (13, 11), (14, 10), (15, 9) merged
            my[0 * kGtVox] = make_float2(a[12], sg[12]);
#pragma unroll
            for (int j = 1; j <= 3; ++j) my[j * kGtVox] = merged(a[12 + j], sg[12 + j], a[12 - j], sg[12 - j]);
        } else if (part == 1) {   // t = 16 .. 24 merged with 8 .. 0; t = 25 .. 31 plain
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int t = 16 + k;
                my[(t - SE) * kGtVox] = t <= 2 * SE ? merged(a[k], sg[k], am[2 * SE - t >= 0 && 2 * SE - t < 9 ? 2 * SE - t : 0],
                                                             sm[2 * SE - t >= 0 && 2 * SE - t < 9 ? 2 * SE - t : 0])
                                                    : make_float2(a[k], sg[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) my[(16 * part + k - SE) * kGtVox] = make_float2(a[k], sg[k]);
        }
        const float log_s_sum = qb::voxel_sum(fmaf(0.5f, dsum, ls)) + (float)T * 0.9189385332046727f;  // + T log sqrt(2 pi)
        __syncthreads();
        if (live) {
            float qv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) qv[i] = q[v * 5 + i];
            const qb::LogitMvn qm = qb::make_mvn(qv);
            const uint64_t vox = (uint64_t)(voxel0 + v);
            float nll_sum = 0.0f, kl_sum = 0.0f;
            int n_kl = 0;
            __builtin_amdgcn_s_setprio(QB_PRIO_LIK);
            const int calls = S > 4 * part ? (S - 4 * part + 15) / 16 : 0;
            const int last = calls > 0 ? S - 4 * (part + 4 * (calls - 1)) : 0;
            const int n_lik = calls > 0 ? 4 * (calls - 1) + (last < 4 ? last : 4) : 0;
            qb::DrawQuad dq;
            uint32_t g = (uint32_t)part;
#pragma unroll 1
            for (int i = 0; i < n_lik; ++i) {
                if ((i & 3) == 0) {
                    dq.load(seed, vox, g, qb::STREAM_LIK);
                    g += QB_LANES_PER_VOXEL;
                }
                float z0, z1, la, lb_;
                dq.next(z0, z1);
                qb::reparam_logits(qm, z0, z1, la, lb_);
                const float sa = qb::sigmoidf_(la), sb = qb::sigmoidf_(lb_);
                const qb::FwdFast fv = qb::fwd_fast_sig(c, sa, sb);
                const float s_se = fmaf(fv.tissue_w, 1.0f, fv.blood_w * qb::exp2f_(fv.ng * bloodB[SE]));   // F(0) = 0
                const float inv_np = qb::rcpf_(s_se + 1e-3f);
                const float lt = qb::log2f_(fv.tissue_w * inv_np), lb = qb::log2f_(fv.blood_w * inv_np);
                const float cg = fminf(sa * (float)NSEG, (float)NSEG - 0.0009765625f);
                const float f = __builtin_amdgcn_fractf(cg);
                const float4* row = gtab + (int)cg;
                float acc;
                {
                    const float2 dd = my[0];
                    const float r = fmaf(-(s_se * inv_np), dd.y, dd.x);
                    acc = r * r;
                }
                struct Stage {
                    float4 kk;
                    float bb;
                    float2 d0;
                };
                auto issue = [&](int e) -> Stage {   // e = t - SE = 1 .. 51
                    Stage st;
                    st.kk = row[(e - 1) * NSEG];
                    st.bb = bloodB[SE + e];
                    st.d0 = my[e * kGtVox];
                    return st;
                };
                auto finish = [&](const Stage& st) {
                    const float F = fmaf(fmaf(fmaf(st.kk.w, f, st.kk.z), f, st.kk.y), f, st.kk.x);
                    const float yh = qb::exp2f_(fmaf(fv.nd, F, lt)) + qb::exp2f_(fmaf(fv.ng, st.bb, lb));
                    const float r = fmaf(-yh, st.d0.y, st.d0.x);
                    acc = fmaf(r, r, acc);
                };
                if (qb_phase_fence()) {   // the tau loop as a basic block of its own (elbo_core.h)
                    constexpr int D = QB_LDS_DEPTH;
                    Stage ring[D + 1];
#pragma unroll
                    for (int k = 0; k < D; ++k) ring[k] = issue(1 + k);
#pragma unroll
                    for (int e = 1; e < NROW; ++e) {
                        if (e + D < NROW) ring[(e - 1 + D) % (D + 1)] = issue(e + D);
                        __builtin_amdgcn_sched_barrier(0);
                        finish(ring[(e - 1) % (D + 1)]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                nll_sum += acc;
            }
            nll_sum = fmaf(0.5f, nll_sum, (float)n_lik * log_s_sum);
            float pv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) pv[i] = prior[v * 5 + i];
            const qb::LogitMvn pm = qb::make_mvn(pv);
            __builtin_amdgcn_s_setprio(QB_PRIO_KL);
            kl_sum = qb::kl_draws_fast(qm, pm, K, nullptr, seed, vox, part, n_kl);
            __builtin_amdgcn_s_setprio(QB_PRIO_AFTER);
            kl_sum = fmaf(0.5f, kl_sum, (float)n_kl * ((pm.s_o + pm.s_d) - (qm.s_o + qm.s_d)));
            const float nll = qb::voxel_sum(nll_sum) / (float)S;
            const float kl = K > 0 ? qb::voxel_sum(kl_sum) / (float)K : 0.0f;
            if (part == 0) {
                const float m = mask ? mask[v] : 1.0f;
                if (nll_kl) nll_kl[v] = make_float2(nll, kl);
                s_nll += nll * m;
                s_kl += m > 0.0f ? kl : 0.0f;
                s_m += m;
            }
        }
    }
    qb::block_partials(red, s_nll, s_kl, s_m, partials);
}

__global__ void reparam_kernel(const float* __restrict__ q, const float2* __restrict__ z,
                               float2* __restrict__ out, int64_t N) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * blockDim.x) {
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) p[k] = q[i * 5 + k];
        const qb::LogitMvn m = qb::make_mvn(p);
        const float2 zz = z[i];
        float a, b, oef, dbv;
        qb::reparam_logits(m, zz.x, zz.y, a, b);
        qb::forward_transform(a, b, oef, dbv);
        out[i] = make_float2(oef, dbv);
    }
}

__global__ void nlogp_kernel(const float2* __restrict__ y, const float* __restrict__ params,
                             float* __restrict__ out, int64_t N) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * blockDim.x) {
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) p[k] = params[i * 5 + k];
        const qb::LogitMvn m = qb::make_mvn(p);
        const float2 yy = y[i];
        out[i] = qb::nlogp(qb::make_obs(yy.x, yy.y), m);
    }
}

// calculate_means(include_r2p=True, return_stds=True) -- model.py:318-343.  Two passes over the
// same counter-generated draws (mean, then biased variance about it), as the reference does.
__global__ void moments_kernel(QbDev c, const float* __restrict__ q, const float* __restrict__ z,
                               int n, uint64_t seed, int64_t voxel0, float* __restrict__ means,
                               float* __restrict__ vars, int64_t N) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * blockDim.x) {
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) p[k] = q[i * 5 + k];
        const qb::LogitMvn m = qb::make_mvn(p);
        const float* zi = z ? z + i * n * 2 : nullptr;
        float mo = 0.0f, md = 0.0f, mr = 0.0f, vo = 0.0f, vd = 0.0f, vr = 0.0f;
        for (int pass = 0; pass < 2; ++pass) {
            float ao = 0.0f, ad = 0.0f, ar = 0.0f;
            for (int j = 0; 2 * j < n; ++j) {
                float zz[4];
                const bool two = 2 * j + 1 < n;
                if (zi) {
                    zz[0] = zi[4 * j];
                    zz[1] = zi[4 * j + 1];
                    zz[2] = two ? zi[4 * j + 2] : 0.0f;
                    zz[3] = two ? zi[4 * j + 3] : 0.0f;
                } else {
                    qb::normals4(seed, (uint64_t)(voxel0 + i), (uint32_t)j, qb::STREAM_MOMENTS, zz);
                }
                for (int d = 0; d < (two ? 2 : 1); ++d) {
                    float a, b, oef, dbv;
                    qb::reparam_logits(m, zz[2 * d], zz[2 * d + 1], a, b);
                    qb::forward_transform(a, b, oef, dbv);
                    const float r2p = (c.dw_coef * oef) * dbv;  // calculate_r2p, model.py:524-525
                    if (pass == 0) {
                        ao += oef;
                        ad += dbv;
                        ar += r2p;
                    } else {
                        ao += (oef - mo) * (oef - mo);
                        ad += (dbv - md) * (dbv - md);
                        ar += (r2p - mr) * (r2p - mr);
                    }
                }
            }
            const float inv = 1.0f / (float)n;
            if (pass == 0) {
                mo = ao * inv;
                md = ad * inv;
                mr = ar * inv;
                if (!vars) break;
            } else {
                vo = ao * inv;
                vd = ad * inv;
                vr = ar * inv;
            }
        }
        means[3 * i] = mo;
        means[3 * i + 1] = md;
        means[3 * i + 2] = mr;
        if (vars) {
            vars[3 * i] = vo;
            vars[3 * i + 1] = vd;
            vars[3 * i + 2] = vr;
        }
    }
}

// The R2' term of synthetic_data_loss (use_r2p_loss, model.py:475-490): n reparameterised draws of
// (OEF, DBV) from the predicted distribution, r_k = dw(OEF_k) DBV_k, a normal fitted to the draws by their
// mean and biased standard deviation (tf.math.reduce_std), and
//   nll = log std + 0.5 ((y - mean) / std)^2                      gaussian_nll, model.py:403-404
// for the true R2' y = y_true[:, 2].  Adds the value to loss_v and scale * d nll / d q to g_q [N][5]:
//   d nll / d r_k = -(y - mean) / (n var) + (1 - (y - mean)^2 / var) (r_k - mean) / (n var),
// chained through calculate_r2p, forward_transform, the reparameterisation and transform_std /
// transform_offdiag.  Three passes over the same counter-generated draws (mean, variance, gradient).
__global__ void r2p_loss_bwd_kernel(QbDev c, const float* __restrict__ y_true, int ldy,
                                    const float* __restrict__ q, const float* __restrict__ z, int n,
                                    uint64_t seed, int64_t voxel0, float scale, float* __restrict__ g_q,
                                    float* __restrict__ loss_v, int64_t N) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * blockDim.x) {
        float p[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) p[k] = q[i * 5 + k];
        const qb::LogitMvn m = qb::make_mvn(p);
        const float* zi = z ? z + i * n * 2 : nullptr;
        const float y = y_true[i * ldy + 2];
        const float inv_n = 1.0f / (float)n;
        float mean = 0.0f, var = 0.0f, A = 0.0f, B = 0.0f;
        float G[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // wrt mu_o, s_o, mu_d, s_d, c
        for (int pass = 0; pass < 3; ++pass) {
            float acc = 0.0f;
            for (int j = 0; 2 * j < n; ++j) {
                float zz[4];
                const bool two = 2 * j + 1 < n;
                if (zi) {
                    zz[0] = zi[4 * j];
                    zz[1] = zi[4 * j + 1];
                    zz[2] = two ? zi[4 * j + 2] : 0.0f;
                    zz[3] = two ? zi[4 * j + 3] : 0.0f;
                } else {
                    qb::normals4(seed, (uint64_t)(voxel0 + i), (uint32_t)j, qb::STREAM_R2P, zz);
                }
                for (int d = 0; d < (two ? 2 : 1); ++d) {
                    const float z0 = zz[2 * d], z1 = zz[2 * d + 1];
                    float a, b, oef, dbv;
                    qb::reparam_logits(m, z0, z1, a, b);
                    qb::forward_transform(a, b, oef, dbv);
                    const float r = (c.dw_coef * oef) * dbv;  // calculate_r2p, model.py:524-525
                    if (pass == 0) {
                        acc += r;
                    } else if (pass == 1) {
                        acc += (r - mean) * (r - mean);
                    } else {
                        const float gr = A + B * (r - mean);
                        // d sigmoid-range / d logit = (x - min) (1 - (x - min) / range)
                        const float uo = oef - QB_MIN_OEF, ud = dbv - QB_MIN_DBV;
                        const float da = gr * c.dw_coef * dbv * uo * (1.0f - uo * (1.0f / QB_OEF_RANGE));
                        const float db = gr * c.dw_coef * oef * ud * (1.0f - ud * (1.0f / QB_DBV_RANGE));
                        G[0] += da;
                        G[1] += da * z0 * m.e_so;
                        G[2] += db;
                        G[3] += db * z1 * m.e_sd;
                        G[4] += db * z0;
                    }
                }
            }
            if (pass == 0) {
                mean = acc * inv_n;
            } else if (pass == 1) {
                var = acc * inv_n;
                const float iv = 1.0f / var, dy = y - mean;
                A = -dy * iv * inv_n;
                B = (1.0f - dy * dy * iv) * iv * inv_n;
                if (loss_v) loss_v[i] += 0.5f * __logf(var) + 0.5f * dy * dy * iv;
                if (!g_q) break;
            }
        }
        if (g_q) {
            const float th1 = (m.s_o + 1.0f) * (1.0f / 3.0f), th3 = (m.s_d + 1.0f) * (1.0f / 3.0f);
            const float th4 = m.c * 7.38905609893065f;
            float* g = g_q + i * 5;
            g[0] += scale * G[0];
            g[1] += scale * G[1] * 3.0f * (1.0f - th1 * th1);                    // transform_std
            g[2] += scale * G[2];
            g[3] += scale * G[3] * 3.0f * (1.0f - th3 * th3);
            g[4] += scale * G[4] * 0.1353352832366127f * (1.0f - th4 * th4);      // transform_offdiag
        }
    }
}

int ew_grid(const qbold_ctx* ctx, int64_t N, int block) {
    int64_t nb = (N + block - 1) / block;
    int64_t cap = (int64_t)ctx->num_cus * 8;
    return (int)(nb < cap ? (nb > 0 ? nb : 1) : cap);
}

}  // namespace

namespace qb {
int elbo_grid(const qbold_ctx* ctx) { return ctx->num_cus * 4; }
// the folded-constant kernels cover the optimal.yaml configuration: full model from the table,
// Gaussian likelihood on linear (not log) data
bool elbo_fast_path(const qbold_ctx* ctx) {
    const QbDev& d = ctx->dev;
    return d.full_model && d.tissue_mode == QBOLD_TISSUE_TABLE && !d.predict_log && !d.use_student_t;
}
}

extern "C" int64_t qbold_elbo_workspace_bytes(const qbold_ctx* ctx) {
    if (!ctx) return QBOLD_ERR_INVALID;
    return (int64_t)sizeof(double) * 3 * (int64_t)qb::elbo_grid(ctx);
}

namespace qb {
// the 64-tau protocol with tau = 0 (up to float32 rounding of the grid) at index 12: see elbo_fwd_launch
bool elbo_logsigma_path(const qbold_ctx* ctx) {
    return ctx->dev.T == 64 && elbo_fast_path(ctx) && ctx->dev.se_idx == 12 && !ctx->dev.multi_norm &&
           !(ctx->kernel_sel & 4) && std::fabs(std::fmaf(12.0f, ctx->dev.tauh_step, ctx->dev.tauh0)) < 1e-6f;
}
}  // namespace qb

// sigma_is_log: `sigma` holds the sigma head before its exp (the wide fused path hands its head over
// that way); built for the protocols that take the LDS-data kernel.
namespace qb {
int elbo_fwd_launch(const qbold_ctx* ctx, const float* x, const float* mask, const float* q,
                    const float* prior, const float* sigma, bool sigma_is_log, const float* zs,
                    const float* zk, int S, int K, uint64_t seed, int64_t voxel0, float* nll_kl,
                    double* sums, void* workspace, int64_t N, hipStream_t s) {
    double* partials = reinterpret_cast<double*>(workspace);
    const int64_t ntile = (N + kVoxPerBlock - 1) / kVoxPerBlock;
    int grid = (int)(ntile < qb::elbo_grid(ctx) ? (ntile > 0 ? ntile : 1) : qb::elbo_grid(ctx));
    const bool lit = ctx->dev.tissue_mode == QBOLD_TISSUE_LITERAL;
    float2* out = reinterpret_cast<float2*>(nll_kl);
#define QB_LAUNCH_ELBO(TT, SE, FAST, LIT)                                                        \
    hipLaunchKernelGGL((elbo_fwd_kernel<TT, SE, FAST, LIT>), dim3(grid), dim3(kBlock), 0, s,      \
                       ctx->dev, ctx->d_tab, x, mask, q, prior, sigma, zs, zk, S, K, seed, voxel0, \
                       out, partials, N)
#define QB_LAUNCH_ELBO_MIR(TT, SE)                                                               \
    hipLaunchKernelGGL((elbo_fwd_kernel<TT, SE, true, false, false, true>), dim3(grid), dim3(kBlock), 0, s, \
                       ctx->dev, ctx->d_tab, x, mask, q, prior, sigma, zs, zk, S, K, seed, voxel0, \
                       out, partials, N)
#define QB_LAUNCH_ELBO_GT(TT, SE)                                                                \
    hipLaunchKernelGGL((elbo_fwd_kernel<TT, SE, true, false, (qb::gtab_segs(TT) > 0)>), dim3(grid), dim3(kBlock), 0, s, \
                       ctx->dev, ctx->d_gtab, x, mask, q, prior, sigma, zs, zk, S, K, seed, voxel0, \
                       out, partials, N)
    const bool gt = ctx->gtab_ok && !(ctx->kernel_sel & 8) && qb::gtab_segs(ctx->dev.T) > 0;
    const bool fast = qb::elbo_fast_path(ctx);
    // long protocols: compile-time tau count and spin-echo index when tau = 0 there (mirrored pairs).  On the
    // float32 grid start + i step the spin-echo tau of config 3 (-0.015 + 12 * 0.00125) is zero only up to
    // rounding: within 1e-6 of a table segment (|dF| < 1e-7, below the table's own float32 rounding) it counts
    // as the spin echo, and tau_{se+j}, tau_{se-j} as a mirrored pair.
    const bool lds64 = qb::elbo_logsigma_path(ctx);
    if (sigma_is_log && !lds64) {
        qb::set_error("elbo_fwd_launch: log-sigma input is built for the 64-tau protocol with tau = 0 at index 12 "
                      "(table mode, Gaussian likelihood, one-image normalisation)");
        return QBOLD_ERR_UNSUPPORTED;
    }
    if (lds64) {
        // elbo_fwd_lds_kernel reads the x and sigma rows as float4: 256-byte rows, so the bases decide
        QB_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(sigma) % 16 == 0,
                   "qbold_elbo_fwd (64-tau protocol): x and sigma must be 16-byte aligned");
        if (ctx->gtab_ok && qb::gtab_segs(64) > 0 && !(ctx->kernel_sel & 8) && !zs && !zk) {
            // per-tau OEF-indexed table, one 1,024-thread workgroup per CU (elbo_fwd_gt64_kernel); explicit normals and
            // QBOLD_KSEL_X_TABLE keep the x-indexed kernel below
            constexpr size_t smem = sizeof(float4) * qb::gtab_taus(64, 12) * qb::gtab_segs(64) + sizeof(float) * 64 +
                                    sizeof(float2) * (64 - 12) * kGtVox + sizeof(double) * 3 * (kGtBlock / 64);
            static_assert(smem <= 160 * 1024, "table + data rows exceed the LDS");
            const int64_t nt = (N + kGtVox - 1) / kGtVox;
            grid = (int)(nt < ctx->num_cus ? (nt > 0 ? nt : 1) : ctx->num_cus);
            auto kern = sigma_is_log ? elbo_fwd_gt64_kernel<64, 12, true> : elbo_fwd_gt64_kernel<64, 12, false>;
            QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kGtBlock), smem, s, ctx->dev, ctx->d_gtab, x, mask, q, prior, sigma,
                               S, K, seed, voxel0, out, partials, N);
            QB_HIP(hipGetLastError());
            hipLaunchKernelGGL(qb::reduce_partials_kernel, dim3(1), dim3(192), 0, s, partials, grid, sums);
            QB_HIP(hipGetLastError());
            return QBOLD_OK;
        }
        const int64_t gt = (N + kLdsVox - 1) / kLdsVox;
        grid = (int)(gt < qb::elbo_grid(ctx) ? (gt > 0 ? gt : 1) : qb::elbo_grid(ctx));
        if (sigma_is_log)
            hipLaunchKernelGGL((elbo_fwd_lds_kernel<64, 12, true>), dim3(grid), dim3(kLdsBlock), 0, s, ctx->dev,
                               ctx->d_tab, x, mask, q, prior, sigma, zs, zk, S, K, seed, voxel0, out, partials, N);
        else
            hipLaunchKernelGGL((elbo_fwd_lds_kernel<64, 12, false>), dim3(grid), dim3(kLdsBlock), 0, s, ctx->dev,
                               ctx->d_tab, x, mask, q, prior, sigma, zs, zk, S, K, seed, voxel0, out, partials, N);
    } else switch (ctx->dev.T) {
        case 11:
            if (fast && ctx->dev.se_idx == 2 && !ctx->dev.multi_norm && gt) QB_LAUNCH_ELBO_GT(11, 2);
            else if (fast && ctx->dev.se_idx == 2 && !ctx->dev.multi_norm && ctx->grid_mirrors) QB_LAUNCH_ELBO_MIR(11, 2);
            else if (fast && ctx->dev.se_idx == 2 && !ctx->dev.multi_norm) QB_LAUNCH_ELBO(11, 2, true, false);
            else if (fast) QB_LAUNCH_ELBO(11, -1, true, false);
            else if (lit && ctx->dev.se_idx == 2 && !ctx->dev.multi_norm) QB_LAUNCH_ELBO(11, 2, false, true);
            else if (lit) QB_LAUNCH_ELBO(11, -1, false, true);
            else QB_LAUNCH_ELBO(11, -1, false, false);
            break;
        case 24:  // the reference's second protocol (signals.py:120-121); se_idx = 7
            if (fast && ctx->dev.se_idx == 7 && !ctx->dev.multi_norm && gt) QB_LAUNCH_ELBO_GT(24, 7);
            else if (fast && ctx->dev.se_idx == 7 && !ctx->dev.multi_norm && ctx->grid_mirrors) QB_LAUNCH_ELBO_MIR(24, 7);
            else if (fast && ctx->dev.se_idx == 7 && !ctx->dev.multi_norm) QB_LAUNCH_ELBO(24, 7, true, false);
            else if (fast) QB_LAUNCH_ELBO(24, -1, true, false);
            else if (lit && ctx->dev.se_idx == 7 && !ctx->dev.multi_norm) QB_LAUNCH_ELBO(24, 7, false, true);
            else if (lit) QB_LAUNCH_ELBO(24, -1, false, true);
            else QB_LAUNCH_ELBO(24, -1, false, false);
            break;
        default: {
            if (!fast) {
                qb::set_error("qbold_elbo_fwd: for T other than 11 / 24 only the optimal.yaml "
                              "configuration (table mode, Gaussian likelihood, linear data) is built");
                return QBOLD_ERR_UNSUPPORTED;
            }
            const int64_t gt = (N + kGenVox - 1) / kGenVox;
            grid = (int)(gt < qb::elbo_grid(ctx) ? (gt > 0 ? gt : 1) : qb::elbo_grid(ctx));
            const size_t smem = sizeof(qb::FwdLds) + sizeof(float) * 2 * QB_MAX_T * kGenVox;
            hipLaunchKernelGGL(elbo_fwd_generic_kernel, dim3(grid), dim3(kGenBlock), smem, s, ctx->dev,
                               ctx->d_tab, x, mask, q, prior, sigma, zs, zk, S, K, seed, voxel0, out,
                               partials, N);
        }
    }
#undef QB_LAUNCH_ELBO
#undef QB_LAUNCH_ELBO_MIR
#undef QB_LAUNCH_ELBO_GT
    QB_HIP(hipGetLastError());
    hipLaunchKernelGGL(qb::reduce_partials_kernel, dim3(1), dim3(192), 0, s, partials, grid, sums);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}
}  // namespace qb

extern "C" int qbold_elbo_fwd(const qbold_ctx* ctx, const float* x, const float* mask, const float* q,
                              const float* prior, const float* sigma, const float* zs,
                              const float* zk, int S, int K, uint64_t seed, int64_t voxel0,
                              float* nll_kl, double* sums, void* workspace, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(N >= 0 && S >= 1 && K >= 0, "qbold_elbo_fwd: need N >= 0, S >= 1, K >= 0");
    QB_REQUIRE(sums && workspace, "qbold_elbo_fwd: null sums/workspace");
    QB_REQUIRE(N == 0 || (x && q && prior && sigma), "qbold_elbo_fwd: null input buffer");
    return qb::elbo_fwd_launch(ctx, x, mask, q, prior, sigma, false, zs, zk, S, K, seed, voxel0, nll_kl, sums,
                               workspace, N, (hipStream_t)stream);
}

extern "C" int qbold_elbo_fwd_logsigma(const qbold_ctx* ctx, const float* x, const float* mask, const float* q,
                                       const float* prior, const float* log_sigma, int S, int K, uint64_t seed,
                                       int64_t voxel0, float* nll_kl, double* sums, void* workspace, int64_t N,
                                       void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(N >= 0 && S >= 1 && K >= 0, "qbold_elbo_fwd_logsigma: need N >= 0, S >= 1, K >= 0");
    QB_REQUIRE(sums && workspace, "qbold_elbo_fwd_logsigma: null sums/workspace");
    QB_REQUIRE(N == 0 || (x && q && prior && log_sigma), "qbold_elbo_fwd_logsigma: null input buffer");
    return qb::elbo_fwd_launch(ctx, x, mask, q, prior, log_sigma, true, nullptr, nullptr, S, K, seed, voxel0, nll_kl,
                               sums, workspace, N, (hipStream_t)stream);
}

// kl_loss for the diagonal family (use_mvg = False, per-voxel prior): model.py:686-716 with
// tfp LogitNormal.kl_divergence = kl_normal_normal of the underlying Gaussians,
//   0.5 ((mu_q - mu_p) / sigma_p)^2 + 0.5 expm1(2 (s_q - s_p)) - (s_q - s_p)   per dimension,
// s = transform_std(raw).  q / prior rows are 5 wide (the fifth, Cholesky, column is unused).
// Analytic, so unlike the sampled KL every q parameter carries gradient.
__global__ __launch_bounds__(256) void kl_diag_kernel(const float* __restrict__ q, const float* __restrict__ prior,
                                                      const float* __restrict__ mask, float* __restrict__ kl_v,
                                                      float* __restrict__ g_q, double* __restrict__ partials,
                                                      int64_t N) {
    __shared__ double red[3 * 4];
    float s_kl = 0.0f, s_m = 0.0f;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < N;
         v += (int64_t)gridDim.x * blockDim.x) {
        float qv[5], pv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            qv[k] = q[v * 5 + k];
            pv[k] = prior[v * 5 + k];
        }
        const float m = mask ? mask[v] : 1.0f;
        const float sqo = qb::transform_std(qv[1]), sqd = qb::transform_std(qv[3]);
        const float spo = qb::transform_std(pv[1]), spd = qb::transform_std(pv[3]);
        const float ipo = __expf(-spo), ipd = __expf(-spd);
        const float do_ = sqo - spo, dd = sqd - spd;
        const float ro = (qv[0] - pv[0]) * ipo, rd = (qv[2] - pv[2]) * ipd;
        const float eo = expm1f(2.0f * do_), ed = expm1f(2.0f * dd);
        const float kl = (0.5f * ro * ro + 0.5f * eo - do_) + (0.5f * rd * rd + 0.5f * ed - dd);
        if (kl_v) kl_v[v] = kl;
        if (m > 0.0f) s_kl += kl;   // model.py:718
        s_m += m;
        if (g_q && m > 0.0f) {
            const float to = (sqo + 1.0f) * (1.0f / 3.0f), td = (sqd + 1.0f) * (1.0f / 3.0f);  // tanh(raw)
            g_q[v * 5 + 0] += ro * ipo;
            g_q[v * 5 + 1] += eo * 3.0f * (1.0f - to * to);   // d/ds_q = exp(2 (s_q - s_p)) - 1
            g_q[v * 5 + 2] += rd * ipd;
            g_q[v * 5 + 3] += ed * 3.0f * (1.0f - td * td);
        }
    }
    qb::block_partials(red, 0.0f, s_kl, s_m, partials);
}

extern "C" int qbold_kl_diag(const qbold_ctx* ctx, const float* q, const float* prior, const float* mask,
                             float* kl_v, float* g_q, double* sums, void* workspace, int64_t N,
                             void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(N >= 0 && sums && workspace, "qbold_kl_diag: null sums/workspace");
    QB_REQUIRE(N == 0 || (q && prior), "qbold_kl_diag: null buffer");
    hipStream_t s = (hipStream_t)stream;
    double* partials = reinterpret_cast<double*>(workspace);
    const int64_t nb = (N + 255) / 256;
    const int grid = (int)(nb < qb::elbo_grid(ctx) ? (nb > 0 ? nb : 1) : qb::elbo_grid(ctx));
    hipLaunchKernelGGL(kl_diag_kernel, dim3(grid), dim3(256), 0, s, q, prior, mask, kl_v, g_q, partials, N);
    QB_HIP(hipGetLastError());
    hipLaunchKernelGGL(qb::reduce_partials_kernel, dim3(1), dim3(192), 0, s, partials, grid, sums);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

// kl_loss against a mixture-of-Gaussians population prior (use_mvg = False, mog_components = M > 1; model.py:666-685):
// one reparameterised draw per dimension, minus the entropy of q, plus the MEAN of the components' Gaussian NLLs.
// comps: device [M][4] raw parameters; z: explicit normals [N][2] (OEF, DBV) or NULL for the Philox stream
// (stream 4, one pair per voxel).
constexpr int kMaxMog = 16;
__global__ __launch_bounds__(256) void kl_mog_kernel(const float* __restrict__ q, const float* __restrict__ comps, int M,
                                                     const float* __restrict__ z, uint64_t seed, int64_t voxel0,
                                                     float* __restrict__ kl_v, int64_t N) {
    __shared__ float cm[kMaxMog][4];   // mean_o, exp(-s_o), mean_d, exp(-s_d)
    __shared__ float cs[kMaxMog];      // s_o + s_d
    if (threadIdx.x < M) {
        const float so = qb::transform_std(comps[4 * threadIdx.x + 1]), sd = qb::transform_std(comps[4 * threadIdx.x + 3]);
        cm[threadIdx.x][0] = comps[4 * threadIdx.x + 0];
        cm[threadIdx.x][1] = __expf(-so);
        cm[threadIdx.x][2] = comps[4 * threadIdx.x + 2];
        cm[threadIdx.x][3] = __expf(-sd);
        cs[threadIdx.x] = so + sd;
    }
    __syncthreads();
    const float inv_m = 1.0f / (float)M;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < N; v += (int64_t)gridDim.x * blockDim.x) {
        float z0, z1;
        if (z) {
            z0 = z[2 * v];
            z1 = z[2 * v + 1];
        } else {
            float zz[4];
            qb::normals4(seed, (uint64_t)(voxel0 + v), 0u, 4u, zz);
            z0 = zz[0];
            z1 = zz[1];
        }
        const float so = qb::transform_std(q[v * 5 + 1]), sd = qb::transform_std(q[v * 5 + 3]);
        const float xo = fmaf(z0, __expf(so), q[v * 5 + 0]), xd = fmaf(z1, __expf(sd), q[v * 5 + 2]);   // :672-675
        float acc = 0.0f;
        for (int c = 0; c < M; ++c) {
            const float ro = (xo - cm[c][0]) * cm[c][1], rd = (xd - cm[c][2]) * cm[c][3];
            acc += cs[c] + 0.5f * fmaf(ro, ro, rd * rd);
        }
        kl_v[v] = fmaf(acc, inv_m, -(so + sd));                                                         // :679-684
    }
}

extern "C" int qbold_kl_mog(const qbold_ctx* ctx, const float* q, const float* comps, int M, const float* z,
                            uint64_t seed, int64_t voxel0, float* kl_v, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && q && comps && kl_v && M >= 1 && M <= kMaxMog, "qbold_kl_mog: bad argument (1 <= M <= 16)");
    hipLaunchKernelGGL(kl_mog_kernel, dim3(ew_grid(ctx, N, 256)), dim3(256), 0, (hipStream_t)stream, q, comps, M, z, seed,
                       voxel0, kl_v, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_reparam(const qbold_ctx* ctx, const float* q, const float* z, float* oef_dbv,
                             int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && q && z && oef_dbv, "qbold_reparam: bad argument");
    hipLaunchKernelGGL(reparam_kernel, dim3(ew_grid(ctx, N, 256)), dim3(256), 0, (hipStream_t)stream,
                       q, reinterpret_cast<const float2*>(z), reinterpret_cast<float2*>(oef_dbv), N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_logit_mvn_nlogp(const qbold_ctx* ctx, const float* y, const float* params,
                                     float* out, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && y && params && out, "qbold_logit_mvn_nlogp: bad argument");
    hipLaunchKernelGGL(nlogp_kernel, dim3(ew_grid(ctx, N, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(y), params, out, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

// squared_whitened_residual (model.py:423-441 = logit_mvn.py:20-38): || L^-1 (obs - mean) ||^2 with the Cholesky
// factor L = [[e^so, 0], [cov, e^sd]] -- a static method of the reference (no distribution object, no context).
__global__ void swr_kernel(const float2* __restrict__ obs, const float2* __restrict__ mean, const float* __restrict__ so,
                           const float* __restrict__ sd, const float* __restrict__ cov, float* __restrict__ out,
                           int64_t N) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const float2 o = obs[i], m = mean[i];
        const float a = so[i], b = sd[i];
        const float inv_tl = __expf(-a), inv_br = __expf(-b);
        const float inv_bl = __expf(-a - b) * cov[i] * -1.0f;            // :431-433
        const float r0 = o.x - m.x, r1 = o.y - m.y;
        const float w0 = r0 * inv_tl, w1 = r1 * inv_br + r0 * inv_bl;    // :435-436
        out[i] = w0 * w0 + w1 * w1;
    }
}

extern "C" int qbold_squared_whitened_residual(const float* obs, const float* mean, const float* oef_log_std,
                                               const float* dbv_log_std, const float* oef_dbv_cov, float* out,
                                               int64_t N, void* stream) {
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && obs && mean && oef_log_std && dbv_log_std && oef_dbv_cov && out,
               "qbold_squared_whitened_residual: bad argument");
    const int64_t nb = (N + 255) / 256;
    hipLaunchKernelGGL(swr_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(obs), reinterpret_cast<const float2*>(mean), oef_log_std,
                       dbv_log_std, oef_dbv_cov, out, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_posterior_moments(const qbold_ctx* ctx, const float* q, const float* z,
                                       int n_samples, uint64_t seed, int64_t voxel0, float* means,
                                       float* vars, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && q && means && n_samples >= 1, "qbold_posterior_moments: bad argument");
    hipLaunchKernelGGL(moments_kernel, dim3(ew_grid(ctx, N, 128)), dim3(128), 0, (hipStream_t)stream,
                       ctx->dev, q, z, n_samples, seed, voxel0, means, vars, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_r2p_loss_bwd(const qbold_ctx* ctx, const float* y_true, int ld_y, const float* q,
                                  const float* z, int n_samples, uint64_t seed, int64_t voxel0, float scale,
                                  float* g_q, float* loss_v, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && y_true && q && ld_y >= 3 && n_samples >= 2 && (g_q || loss_v),
               "qbold_r2p_loss_bwd: bad argument (y_true rows need the R2' column, at least two draws)");
    hipLaunchKernelGGL(r2p_loss_bwd_kernel, dim3(ew_grid(ctx, N, 128)), dim3(128), 0, (hipStream_t)stream,
                       ctx->dev, y_true, ld_y, q, z, n_samples, seed, voxel0, scale, g_q, loss_v, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

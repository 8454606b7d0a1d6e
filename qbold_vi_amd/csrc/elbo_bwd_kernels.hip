// elbo_bwd_kernels.hip -- gradient of the per-voxel negative ELBO with respect to the encoder's
// head outputs (posterior parameters q[5] and log sigma[T]).
//
// This is the adjoint of build_fine_tuner's sampling path + fine_tune_loss_fn + kl_loss
// (model.py:239-286, 527-568, 592-610, 654-665) that Keras/TensorFlow autodiff produces in the
// reference's fit loop (train.py:315-376): reparameterisation gradients through forward_transform
// and the forward signal model (tissue-integral slope = the J1 Simpson sum, see QbDev::dF_node0),
// the Gaussian NLL with heteroscedastic sigma, and the Monte-Carlo KL whose q-parameters are
// stop-gradient inside log q (model.py:596) so that only the samples carry gradient.
//
// Outputs are per-voxel and UNNORMALISED by the mask sum: g_head[v] = m_v * d(nll_v)/d(.) +
// [m_v > 0] * d(kl_v)/d(.), so the caller divides the weight gradient by sum(m) once.
// Same lane mapping and Philox stream as the forward kernels, so loss values are identical.
#include "elbo_core.h"
#include "qbold_ctx.h"
#ifndef QB_BWD_WHITENED
#define QB_BWD_WHITENED 1
#endif

namespace qb {
bool elbo_fast_path(const qbold_ctx* ctx);
int elbo_grid(const qbold_ctx* ctx);
}

namespace {

constexpr int kBlock = 256;

struct FwdGrad {
    float s, ds_doef, ds_ddbv;
};

// signal and its partials at tau index t (full model, table mode)
__device__ __forceinline__ FwdGrad fwd_signal_grad(const qb::FwdLds* L, const QbDev& c,
                                                   const qb::FwdFast& v, float oef, float dbv, int t) {
    const float us = fmaf((float)t, v.ub, v.ua);  // signed table coordinate
    const float u = fabsf(us);
    const int i = min((int)u, QB_TAB_SEG - 1);
    const float f = u - (float)i;
    const float4 k = L->tab[i];
    const float F = fmaf(fmaf(fmaf(k.w, f, k.z), f, k.y), f, k.x);
    // dF/dx at |x| (x = u / tab_inv_h), plus node 0's slope; d|x|/doef = |x| / oef
    const float ax = u * (1.0f / c.tab_inv_h);
    const float dF = fmaf(fmaf(3.0f * k.w, f, 2.0f * k.z), f, k.y) * c.tab_inv_h + c.dF_node0 * ax;
    const float e1 = qb::exp2f_(v.nd * F);
    const float e2 = qb::exp2f_(v.ng * c.blood_B[t]);
    const float tissue = v.tissue_w * e1, blood = v.blood_w * e2;
    FwdGrad g;
    g.s = tissue + blood;
    const float inv_oef = qb::rcpf_(oef);
    // tissue: exp(-dbv F(x)), x proportional to oef;  blood: exp(-g B), g proportional to oef^2
    g.ds_doef = -dbv * dF * ax * inv_oef * tissue +
                (2.0f * QB_LN2) * v.ng * c.blood_B[t] * inv_oef * blood;
    // weights: tissue_w = (1 - bw) C1, blood_w = bw C2, bw = m_bld_nb dbv (or dbv without blood)
    const float dbw = c.include_blood ? c.m_bld_nb : 1.0f;
    g.ds_ddbv = -F * tissue - dbw * c.e_te_r2t * e1 + (c.include_blood ? dbw * c.e_r2b_te * e2 : 0.0f);
    return g;
}

// LPV = lanes per voxel.  4 is the forward kernels' mapping (a voxel's draws dealt to four lanes, lane groups 16
// apart): right for the evaluation's S = 32 draws.  Training runs the reference's defaults S = 1, K = 70: with four
// lanes per voxel the one likelihood draw keeps one lane in four busy while all four repeat the voxel's loads and
// its per-voxel preparation, 3,000 vector instructions per 16 voxels.  LPV = 1 (chosen for S <= 2) gives every
// lane its own voxel and walks the draws in order: 3,700 per 64.  The draws themselves are keyed by (seed, voxel,
// draw index), not by lane: the two mappings differ in summation order only.
template <int T, int SE, int LPV>
__global__ __launch_bounds__(kBlock) void elbo_bwd_kernel(
    QbDev c, const float4* __restrict__ g_tab, const float* __restrict__ x,
    const float* __restrict__ mask, const float* __restrict__ q, const float* __restrict__ prior,
    const float* __restrict__ log_sigma, int S, int K, uint64_t seed, int64_t voxel0,
    float* __restrict__ g_q, float* __restrict__ g_ls, float2* __restrict__ nll_kl,
    double* __restrict__ partials, int64_t N) {
    __shared__ qb::FwdLds L;
    __shared__ double red[3 * (kBlock / 64)];
    qb::fwd_lds_fill(&L, g_tab, false);
    __syncthreads();

    constexpr int kVoxPerBlock = kBlock / LPV, kVoxPerWave = 64 / LPV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = LPV == 1 ? 0 : lane >> 4;
    auto voxel_sum = [](float x) { return LPV == 1 ? x : qb::voxel_sum(x); };
    float s_nll = 0.0f, s_kl = 0.0f, s_m = 0.0f;
    const int64_t ntile = (N + kVoxPerBlock - 1) / kVoxPerBlock;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t v = tile * kVoxPerBlock + wave * kVoxPerWave + (LPV == 1 ? lane : lane & 15);
        if (v < N) {
            float xv[T], lsv[T], qv[5], pv[5];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                xv[t] = x[v * T + t];
                lsv[t] = log_sigma[v * T + t];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                qv[i] = q[v * 5 + i];
                pv[i] = prior[v * 5 + i];
            }
            const float m = mask ? mask[v] : 1.0f;
            qb::VoxelLik<T> lik;
            qb::prepare_lik<T, SE, true>(c, xv, lsv, m, lik);
            const qb::LogitMvn qm = qb::make_mvn(qv), pm = qb::make_mvn(pv);
            const uint64_t vox = (uint64_t)(voxel0 + v);

            // accumulators: gradient wrt the logit-space sample parameters and log sigma
            float g_mu_o = 0.0f, g_so = 0.0f, g_mu_d = 0.0f, g_sd = 0.0f, g_c = 0.0f;
            float gls[T];
#pragma unroll
            for (int t = 0; t < T; ++t) gls[t] = 0.0f;
            float nll_sum = 0.0f, kl_sum = 0.0f;
            int n_lik = 0, n_kl = 0;

            // ---- likelihood draws --------------------------------------------------------
            for (int g = part; 4 * g < S; g += LPV) {
                const int cnt = S - 4 * g < 4 ? S - 4 * g : 4;
                n_lik += cnt;
                qb::DrawQuad dq;
                dq.load(seed, vox, (uint32_t)g, qb::STREAM_LIK);
#pragma unroll 1
                for (int d = 0; d < cnt; ++d) {
                    float z0, z1;
                    dq.next(z0, z1);
                    float a, b;
                    qb::reparam_logits(qm, z0, z1, a, b);
                    const float sa = qb::sigmoidf_(a), sb = qb::sigmoidf_(b);
                    const float oef = sa * QB_OEF_RANGE + QB_MIN_OEF;
                    const float dbv = sb * QB_DBV_RANGE + QB_MIN_DBV;
                    const qb::FwdFast fv = qb::fwd_fast(c, oef, dbv);
                    // pass 1: signals, residuals, NLL, d nll / d yhat
                    float sig[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) sig[t] = qb::fwd_signal_fast(&L, c, fv, t);
                    const float inv_np = qb::rcpf_(qb::se_norm<T, SE>(c, sig));
                    float acc = 0.0f, a1 = 0.0f;
                    float gy[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        float yp = sig[t] * inv_np, dyp = 1.0f;   // yhat_t and d yp / d yhat_t
                        if (c.predict_log) {                      // model.py:547-549
                            dyp = m > 0.0f ? qb::rcpf_(yp) : 0.0f;
                            yp = m > 0.0f ? __logf(yp) : 0.0f;
                        }
                        const float r = (lik.yt[t] - yp) * lik.inv_s[t];
                        float dr = r;                             // d nll_t / d r
                        if (c.use_student_t) {                    // -StudentT(df, 0, sigma).log_prob, :557-559
                            const float w = (c.st_df + 1.0f) * qb::rcpf_(fmaf(r, r, c.st_df));
                            acc += (c.st_df + 1.0f) * log1pf(r * r * qb::rcpf_(c.st_df)) - 2.0f * c.st_const;
                            dr = w * r;
                        } else {
                            acc = fmaf(r, r, acc);
                        }
                        gls[t] += 1.0f - dr * r;                  // d/d log sigma_t of log sigma_t + nll_t(r)
                        gy[t] = -dr * lik.inv_s[t] * dyp;         // d nll / d yhat_t
                        a1 = fmaf(gy[t], sig[t], a1);
                    }
                    nll_sum += acc;
                    a1 *= inv_np * inv_np;               // yhat_t = sig_t / (norm + 1e-3)
                    // pass 2: chain through the forward model
                    float g_oef = 0.0f, g_dbv = 0.0f;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        float gs = gy[t] * inv_np;
                        if (SE >= 0) {
                            if (t == SE) gs -= a1;
                        } else if (c.multi_norm) {
                            if (t >= c.se_idx - 1 && t <= c.se_idx + 1) gs -= a1 * (1.0f / 3.0f);
                        } else if (t == c.se_idx) {
                            gs -= a1;
                        }
                        const FwdGrad fg = fwd_signal_grad(&L, c, fv, oef, dbv, t);
                        g_oef = fmaf(gs, fg.ds_doef, g_oef);
                        g_dbv = fmaf(gs, fg.ds_ddbv, g_dbv);
                    }
                    const float ga = g_oef * QB_OEF_RANGE * sa * (1.0f - sa);  // forward_transform
                    const float gb = g_dbv * QB_DBV_RANGE * sb * (1.0f - sb);
                    g_mu_o += ga;
                    g_so = fmaf(ga, z0 * qm.e_so, g_so);
                    g_mu_d += gb;
                    g_c = fmaf(gb, z0, g_c);
                    g_sd = fmaf(gb, z1 * qm.e_sd, g_sd);
                }
            }
            const float inv_S = 1.0f / (float)S;
            const float wn = m * inv_S;  // weight of one likelihood draw in m_v * nll_v
            g_mu_o *= wn; g_so *= wn; g_mu_d *= wn; g_sd *= wn; g_c *= wn;
#pragma unroll
            for (int t = 0; t < T; ++t) gls[t] *= wn;
            nll_sum = fmaf(0.5f, nll_sum, (float)n_lik * lik.log_s_sum);

            // ---- KL draws ------------------------------------------------------------------
            float k_mu_o = 0.0f, k_so = 0.0f, k_mu_d = 0.0f, k_sd = 0.0f, k_c = 0.0f;
            // Whitened form (elbo_core.h, kl_draws_fast): under q the whitened residual of a draw is z itself, under
            // the prior it is w = d + M z with d, M fixed per voxel.  Both the value, sum |w|^2 - |z|^2, and the
            // gradient through the sample (q stop-gradient inside log q, model.py:596) --
            //   g_l = L_p^-T w - L_q^-T z,  d/d mu += g_l,  d/d s_o += g_a z0 e^so,  d/d c += g_b z0,  d/d s_d += g_b z1 e^sd
            // -- are then linear / quadratic in (1, z0, z1): the loop only gathers the draws' five second moments (five
            // instructions per draw besides Philox, instead of thirty-five) and the sums are assembled once per voxel.
            // Valid while the logit clip cannot bind (the same per-wave bound as the forward kernels).
            constexpr float kZMax = QB_Z_MAX;
            const float reach = fmaxf(fabsf(qm.mu_o) + kZMax * qm.e_so, fabsf(qm.mu_d) + kZMax * (fabsf(qm.c) + qm.e_sd));
            if (QB_BWD_WHITENED && __all(reach < QB_LOGIT_CLIP)) {
                float s0 = 0.0f, s1 = 0.0f, s00 = 0.0f, s11 = 0.0f, s01 = 0.0f;
                for (int g = part; 4 * g < K; g += LPV) {
                    float z[8];
                    const int cnt = K - 4 * g < 4 ? K - 4 * g : 4;
                    qb::normals8_unscaled(seed, vox, (uint32_t)g, qb::STREAM_KL, cnt, z);   // beyond K: z = 0, adds nothing to any moment
                    n_kl += cnt;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float z0 = z[2 * d], z1 = z[2 * d + 1];
                        s0 += z0;
                        s1 += z1;
                        s00 = fmaf(z0, z0, s00);
                        s11 = fmaf(z1, z1, s11);
                        s01 = fmaf(z0, z1, s01);
                    }
                }
                // the normals' common factor sqrt(2 ln 2) (normals8_unscaled), applied to the moments
                s0 *= QB_BM_K;
                s1 *= QB_BM_K;
                s00 *= QB_BM_K * QB_BM_K;
                s11 *= QB_BM_K * QB_BM_K;
                s01 *= QB_BM_K * QB_BM_K;
                const float nk = (float)n_kl;
                const float dmu_o = qm.mu_o - pm.mu_o, dmu_d = qm.mu_d - pm.mu_d;
                const float d0 = dmu_o * pm.i_so, m00 = qm.e_so * pm.i_so;
                const float d1 = fmaf(dmu_d, pm.i_sd, dmu_o * pm.i_bl);
                const float m10 = fmaf(qm.c, pm.i_sd, qm.e_so * pm.i_bl), m11 = qm.e_sd * pm.i_sd;
                // sum |w|^2 with w0 = d0 + m00 z0, w1 = d1 + m10 z0 + m11 z1
                const float sw0 = nk * d0 * d0 + 2.0f * d0 * m00 * s0 + m00 * m00 * s00;
                const float sw1 = nk * d1 * d1 + 2.0f * d1 * (m10 * s0 + m11 * s1) + m10 * m10 * s00 +
                                  2.0f * m10 * m11 * s01 + m11 * m11 * s11;
                kl_sum = (sw0 + sw1) - (s00 + s11);
                // g_a = A0 + A1 z0 + A2 z1,  g_b = B0 + B1 z0 + B2 z1
                const float A0 = d0 * pm.i_so + d1 * pm.i_bl;
                const float A1 = m00 * pm.i_so + m10 * pm.i_bl - qm.i_so;
                const float A2 = m11 * pm.i_bl - qm.i_bl;
                const float B0 = d1 * pm.i_sd, B1 = m10 * pm.i_sd, B2 = m11 * pm.i_sd - qm.i_sd;
                k_mu_o = A0 * nk + A1 * s0 + A2 * s1;
                k_so = (A0 * s0 + A1 * s00 + A2 * s01) * qm.e_so;
                k_mu_d = B0 * nk + B1 * s0 + B2 * s1;
                k_c = B0 * s0 + B1 * s00 + B2 * s01;
                k_sd = (B0 * s1 + B1 * s01 + B2 * s11) * qm.e_sd;
            } else
            for (int g = part; 4 * g < K; g += LPV) {
                const int cnt = K - 4 * g < 4 ? K - 4 * g : 4;
                n_kl += cnt;
                qb::DrawQuad dq;
                dq.load(seed, vox, (uint32_t)g, qb::STREAM_KL);
#pragma unroll 1
                for (int d = 0; d < cnt; ++d) {
                    float z0, z1;
                    dq.next(z0, z1);
                    float a, b;
                    qb::reparam_logits(qm, z0, z1, a, b);
                    const float l0 = qb::clampf_(a, -QB_LOGIT_CLIP, QB_LOGIT_CLIP);
                    const float l1 = qb::clampf_(b, -QB_LOGIT_CLIP, QB_LOGIT_CLIP);
                    const float rq0 = l0 - qm.mu_o, rq1 = l1 - qm.mu_d;
                    const float rp0 = l0 - pm.mu_o, rp1 = l1 - pm.mu_d;
                    const float wq0 = rq0 * qm.i_so, wq1 = fmaf(rq1, qm.i_sd, rq0 * qm.i_bl);
                    const float wp0 = rp0 * pm.i_so, wp1 = fmaf(rp1, pm.i_sd, rp0 * pm.i_bl);
                    kl_sum += fmaf(wp0, wp0, wp1 * wp1) - fmaf(wq0, wq0, wq1 * wq1);
                    // d/dl of 0.5 (swr_p - swr_q); the clip passes gradient
                    // (tfp clip_by_value_preserve_gradient, model.py:395)
                    const float ga = (wp0 * pm.i_so + wp1 * pm.i_bl) - (wq0 * qm.i_so + wq1 * qm.i_bl);
                    const float gb = wp1 * pm.i_sd - wq1 * qm.i_sd;
                    k_mu_o += ga;
                    k_so = fmaf(ga, z0 * qm.e_so, k_so);
                    k_mu_d += gb;
                    k_c = fmaf(gb, z0, k_c);
                    k_sd = fmaf(gb, z1 * qm.e_sd, k_sd);
                }
            }
            kl_sum = fmaf(0.5f, kl_sum, (float)n_kl * ((pm.s_o + pm.s_d) - (qm.s_o + qm.s_d)));
            if (K > 0) {
                const float wk = (m > 0.0f ? 1.0f : 0.0f) / (float)K;
                g_mu_o = fmaf(wk, k_mu_o, g_mu_o);
                g_so = fmaf(wk, k_so, g_so);
                g_mu_d = fmaf(wk, k_mu_d, g_mu_d);
                g_sd = fmaf(wk, k_sd, g_sd);
                g_c = fmaf(wk, k_c, g_c);
            }
            // combine the lanes of the voxel
            g_mu_o = voxel_sum(g_mu_o);
            g_so = voxel_sum(g_so);
            g_mu_d = voxel_sum(g_mu_d);
            g_sd = voxel_sum(g_sd);
            g_c = voxel_sum(g_c);
#pragma unroll
            for (int t = 0; t < T; ++t) gls[t] = voxel_sum(gls[t]);
            const float nll = voxel_sum(nll_sum) * inv_S;
            const float kl = K > 0 ? voxel_sum(kl_sum) / (float)K : 0.0f;
            if (part == 0) {
                // transform_std / transform_offdiag (model.py:288-294): s = 3 tanh(raw) - 1,
                // c = tanh(raw) e^-2.  NB the -(s_o + s_d)_q term of the KL is stop-gradient.
                const float th1 = (qm.s_o + 1.0f) * (1.0f / 3.0f), th3 = (qm.s_d + 1.0f) * (1.0f / 3.0f);
                const float th4 = qm.c * 7.38905609893065f;
                g_q[v * 5 + 0] = g_mu_o;
                g_q[v * 5 + 1] = g_so * 3.0f * (1.0f - th1 * th1);
                g_q[v * 5 + 2] = g_mu_d;
                g_q[v * 5 + 3] = g_sd * 3.0f * (1.0f - th3 * th3);
                g_q[v * 5 + 4] = g_c * 0.1353352832366127f * (1.0f - th4 * th4);
                if (nll_kl) nll_kl[v] = make_float2(nll, kl);
                s_nll += nll * m;
                s_kl += m > 0.0f ? kl : 0.0f;
                s_m += m;
            }
            if (part == (LPV == 1 ? 0 : 1)) {
#pragma unroll
                for (int t = 0; t < T; ++t) g_ls[v * T + t] = gls[t];
            }
        }
    }
    qb::block_partials(red, s_nll, s_kl, s_m, partials);
}

}  // namespace

extern "C" int qbold_elbo_bwd(const qbold_ctx* ctx, const float* x, const float* mask, const float* q,
                              const float* prior, const float* log_sigma, int S, int K, uint64_t seed,
                              int64_t voxel0, float* g_q, float* g_log_sigma, float* nll_kl,
                              double* sums, void* workspace, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(N >= 0 && S >= 1 && K >= 0, "qbold_elbo_bwd: need N >= 0, S >= 1, K >= 0");
    QB_REQUIRE(sums && workspace, "qbold_elbo_bwd: null sums/workspace");
    QB_REQUIRE(N == 0 || (x && q && prior && log_sigma && g_q && g_log_sigma),
               "qbold_elbo_bwd: null buffer");
    if (!(ctx->dev.full_model && ctx->dev.tissue_mode == QBOLD_TISSUE_TABLE)) {
        qb::set_error("qbold_elbo_bwd: gradients are built for the full signal model in table mode "
                      "(Gaussian or Student-t likelihood, linear or log data, either normalisation)");
        return QBOLD_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    double* partials = reinterpret_cast<double*>(workspace);
    const int lpv = S <= 2 ? 1 : QB_LANES_PER_VOXEL;   // see elbo_bwd_kernel
    const int vpb = kBlock / lpv;
    const int64_t ntile = (N + vpb - 1) / vpb;
    const int grid = (int)(ntile < qb::elbo_grid(ctx) ? (ntile > 0 ? ntile : 1) : qb::elbo_grid(ctx));
    float2* out = reinterpret_cast<float2*>(nll_kl);
#define QB_LAUNCH_BWD(TT, SEC)                                                                                          \
    do {                                                                                                                \
        if (lpv == 1)                                                                                                   \
            hipLaunchKernelGGL((elbo_bwd_kernel<TT, SEC, 1>), dim3(grid), dim3(kBlock), 0, s, ctx->dev, ctx->d_tab, x,   \
                               mask, q, prior, log_sigma, S, K, seed, voxel0, g_q, g_log_sigma, out, partials, N);      \
        else                                                                                                            \
            hipLaunchKernelGGL((elbo_bwd_kernel<TT, SEC, QB_LANES_PER_VOXEL>), dim3(grid), dim3(kBlock), 0, s, ctx->dev, \
                               ctx->d_tab, x, mask, q, prior, log_sigma, S, K, seed, voxel0, g_q, g_log_sigma, out,     \
                               partials, N);                                                                            \
    } while (0)
    if (ctx->dev.T == 24) {
        QB_LAUNCH_BWD(24, -1);
    } else if (ctx->dev.T != 11) {
        qb::set_error("qbold_elbo_bwd: kernels are built for T = 11 or 24 taus");
        return QBOLD_ERR_UNSUPPORTED;
    } else if (ctx->dev.se_idx == 2 && !ctx->dev.multi_norm) {
        QB_LAUNCH_BWD(11, 2);
    } else {
        QB_LAUNCH_BWD(11, -1);
    }
#undef QB_LAUNCH_BWD
    QB_HIP(hipGetLastError());
    hipLaunchKernelGGL(qb::reduce_partials_kernel, dim3(1), dim3(192), 0, s, partials, grid, sums);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

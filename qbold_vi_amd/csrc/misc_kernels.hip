// misc_kernels.hip -- the small per-voxel pieces of the reference's API surface that are not
// worth a fused kernel of their own: normalise_data, the parameter transforms, the stand-alone
// NLL and KL terms, the Philox normal stream, and the synthetic-data noise model.
#include "elbo_core.h"
#include "qbold_ctx.h"

namespace {

int ew_grid(const qbold_ctx* ctx, int64_t n, int block) {
    int64_t nb = (n + block - 1) / block;
    int64_t cap = (int64_t)ctx->num_cus * 8;
    return (int)(nb < cap ? (nb > 0 ? nb : 1) : cap);
}

// EncoderTrainer.normalise_data -- model.py:97-113
__global__ void normalise_kernel(QbDev c, const float* __restrict__ x, float* __restrict__ out,
                                 int64_t N) {
    const int T = c.T, se = c.se_idx;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < N;
         v += (int64_t)gridDim.x * blockDim.x) {
        const float* xv = x + v * T;
        float den;
        if (c.multi_norm)
            den = (qb::clampf_(xv[se - 1], 1e-2f, 1e8f) + qb::clampf_(xv[se], 1e-2f, 1e8f) +
                   qb::clampf_(xv[se + 1], 1e-2f, 1e8f)) / 3.0f;
        else
            den = qb::clampf_(xv[se], 1e-2f, 1e8f);
        for (int t = 0; t < T; ++t) out[v * T + t] = logf(qb::clampf_(xv[t], 1e-2f, 1e8f) / den);
    }
}

// transform_std / transform_offdiag / inv_transform_std (model.py:288-297) and
// forward_transform / backwards_transform on interleaved (OEF, DBV) pairs (model.py:299-316)
__global__ void transform_kernel(int op, const float* __restrict__ in, float* __restrict__ out,
                                 int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float v = in[i];
        const bool is_dbv = i & 1;
        float r;
        switch (op) {
            case QBOLD_TRANSFORM_STD: r = tanhf(v) * 3.0f - 1.0f; break;
            case QBOLD_TRANSFORM_OFFDIAG: r = tanhf(v) * 0.1353352832366127f; break;
            case QBOLD_INV_TRANSFORM_STD: r = atanhf((v + 1.0f) / 3.0f); break;
            case QBOLD_FORWARD_TRANSFORM:
                r = is_dbv ? (1.0f / (1.0f + expf(-v))) * QB_DBV_RANGE + QB_MIN_DBV
                           : (1.0f / (1.0f + expf(-v))) * QB_OEF_RANGE + QB_MIN_OEF;
                break;
            case QBOLD_BACKWARDS_TRANSFORM:
            case QBOLD_BACKWARDS_TRANSFORM_LOGIT: {
                r = is_dbv ? (v - QB_MIN_DBV) / QB_DBV_RANGE : (v - QB_MIN_OEF) / QB_OEF_RANGE;
                if (op == QBOLD_BACKWARDS_TRANSFORM_LOGIT) r = logf(r / (1.0f - r));
                break;
            }
            case QBOLD_EXP: r = expf(v); break;
            default: r = v;
        }
        out[i] = r;
    }
}

// fine_tune_loss_fn(return_mean=False) before the mask multiply -- model.py:527-563:
// x [N][T] data, pred [N*S][T] predicted signals, sigma [N*S][T]; row j of pred belongs to voxel
// j % N (the reference tiles y_true S times along the batch axis, model.py:529).
__global__ void nll_kernel(QbDev c, const float* __restrict__ x, const float* __restrict__ mask,
                           const float* __restrict__ pred, const float* __restrict__ sigma,
                           float* __restrict__ nll, int64_t N, int64_t rows) {
    const int T = c.T, se = c.se_idx;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < rows;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = j % N;
        const float* xv = x + v * T;
        const float* pv = pred + j * T;
        const float* sv = sigma + j * T;
        const float m = mask ? mask[v] : 1.0f;
        float nt, np_;
        if (c.multi_norm) {
            nt = (xv[se - 1] + xv[se] + xv[se + 1]) / 3.0f + 1e-3f;
            np_ = (pv[se - 1] + pv[se] + pv[se + 1]) / 3.0f + 1e-3f;
        } else {
            nt = xv[se] + 1e-3f;
            np_ = pv[se] + 1e-3f;
        }
        float acc = 0.0f;
        for (int t = 0; t < T; ++t) {
            float yt = xv[t] / nt, yp = pv[t] / np_;
            if (c.predict_log) {
                yt = m > 0.0f ? logf(yt) : 0.0f;
                yp = m > 0.0f ? logf(yp) : 0.0f;
            }
            const float r = (yt - yp) / sv[t];
            if (c.use_student_t)
                acc += -(c.st_const - logf(sv[t]) - 0.5f * (c.st_df + 1.0f) * log1pf(r * r / c.st_df));
            else
                acc += logf(sv[t]) + 0.9189385332046727f + 0.5f * r * r;
        }
        nll[j] = acc;
    }
}

// mvg_kl_samples (model.py:592-610) per voxel; zk explicit [N][K][2] or the Philox KL stream
__global__ void kl_samples_kernel(const float* __restrict__ q, const float* __restrict__ prior,
                                  const float* __restrict__ zk, int K, uint64_t seed, int64_t voxel0,
                                  float* __restrict__ kl, int64_t N) {
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < N;
         v += (int64_t)gridDim.x * blockDim.x) {
        float qv[5], pv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            qv[k] = q[v * 5 + k];
            pv[k] = prior[v * 5 + k];
        }
        const qb::LogitMvn qm = qb::make_mvn(qv), pm = qb::make_mvn(pv);
        const float* zv = zk ? zk + v * K * 2 : nullptr;
        float acc = 0.0f;
        for (int j = 0; 2 * j < K; ++j) {
            float z[4];
            const bool two = 2 * j + 1 < K;
            if (zv) {
                z[0] = zv[4 * j];
                z[1] = zv[4 * j + 1];
                z[2] = two ? zv[4 * j + 2] : 0.0f;
                z[3] = two ? zv[4 * j + 3] : 0.0f;
            } else {
                qb::normals4(seed, (uint64_t)(voxel0 + v), (uint32_t)j, qb::STREAM_KL, z);
            }
            for (int d = 0; d < (two ? 2 : 1); ++d) {
                float a, b, oef, dbv;
                qb::reparam_logits(qm, z[2 * d], z[2 * d + 1], a, b);
                qb::forward_transform(a, b, oef, dbv);
                const qb::LogitObs o = qb::make_obs(oef, dbv);
                acc += qb::nlogp(o, pm) - qb::nlogp(o, qm);
            }
        }
        kl[v] = acc / (float)K;
    }
}

// mvg_kl closed form (use_population_prior = False) -- model.py:612-652
__global__ void kl_closed_kernel(const float* __restrict__ q, const float* __restrict__ prior,
                                 float* __restrict__ kl, int64_t N) {
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < N;
         v += (int64_t)gridDim.x * blockDim.x) {
        float qv[5], pv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            qv[k] = q[v * 5 + k];
            pv[k] = prior[v * 5 + k];
        }
        const qb::LogitMvn Q = qb::make_mvn(qv), P = qb::make_mvn(pv);
        const float r0 = P.mu_o - Q.mu_o, r1 = P.mu_d - Q.mu_d;  // :633
        const float w0 = r0 * P.i_so, w1 = r1 * P.i_sd + r0 * P.i_bl;
        const float sq = w0 * w0 + w1 * w1;
        const float det_term = 2.0f * (P.s_o + P.s_d) - 2.0f * (Q.s_o + Q.s_d);  // :628-635
        const float inv_p_od = P.i_so * P.c * P.i_sd * -1.0f;                     // :639
        const float inv_pcov_tl = P.i_so * P.i_so;
        const float inv_pcov_br = inv_p_od * inv_p_od + P.i_sd * P.i_sd;
        const float inv_pcov_od = P.i_so * inv_p_od;
        const float q_tl = Q.e_so * Q.e_so;
        const float q_br = Q.e_sd * Q.e_sd + Q.c * Q.c;
        const float q_od = Q.c * Q.e_so;
        const float trace = inv_pcov_tl * q_tl + inv_pcov_od * q_od + inv_pcov_od * q_od +
                            q_br * inv_pcov_br;  // :648
        kl[v] = 0.5f * (trace + sq + det_term - 2.0f);  // :651
    }
}

// z [N][n][2]: the counter-based normal stream the fused kernels consume
__global__ void normals_kernel(uint64_t seed, uint32_t stream, int64_t voxel0, int n,
                               float* __restrict__ z, int64_t N) {
    const int pairs = (n + 1) / 2;
    const int64_t total = N * pairs;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < total;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = k / pairs;
        const int j = (int)(k % pairs);
        float zz[4];
        qb::normals4(seed, (uint64_t)(voxel0 + v), (uint32_t)j, stream, zz);
        float* o = z + (v * n + 2 * j) * 2;
        o[0] = zz[0];
        o[1] = zz[1];
        if (2 * j + 1 < n) {
            o[2] = zz[2];
            o[3] = zz[3];
        }
    }
}

// column sums of signal [V][T] in double: partial[blockIdx][t]
__global__ void colsum_kernel(const float* __restrict__ s, int T, int64_t V,
                              double* __restrict__ partial) {
    extern __shared__ double sh[];  // [blockDim.x]
    for (int t = 0; t < T; ++t) {
        double a = 0.0;
        for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V;
             v += (int64_t)gridDim.x * blockDim.x)
            a += (double)s[v * T + t];
        sh[threadIdx.x] = a;
        __syncthreads();
        for (int o = blockDim.x / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * T + t] = sh[0];
        __syncthreads();
    }
}

// Noise model of SignalGenerationLayer.call -- signals.py:116-128:
//   snr[v][t] = U(50,120)[v] * norm_snr[t];  std[v][t] = mean_v(signal[.][t]) / snr[v][t];
//   signal += N(0,1) * std.   Randomness: Philox stream 3, counter (voxel, draw pair).
__global__ void add_noise_kernel(float* __restrict__ s, int T, int64_t V,
                                 const double* __restrict__ partial, int nblocks,
                                 const float* __restrict__ norm_snr, float snr_lo, float snr_hi,
                                 uint64_t seed, int64_t voxel0) {
    __shared__ float mean[QB_MAX_T];
    if ((int)threadIdx.x < T) {
        double a = 0.0;
        for (int b = 0; b < nblocks; ++b) a += partial[(int64_t)b * T + threadIdx.x];
        mean[threadIdx.x] = (float)(a / (double)V);
    }
    __syncthreads();
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V;
         v += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t vox = (uint64_t)(voxel0 + v);
        const uint4 u = qb::philox4x32_10(make_uint4((uint32_t)vox, (uint32_t)(vox >> 32), 0xFFFFFFFFu, 3u),
                                          make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const float uni = ((float)(u.x >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float snr_v = snr_lo + (snr_hi - snr_lo) * uni;
        for (int j = 0; 4 * j < T; ++j) {
            float z[4];
            qb::normals4(seed, vox, (uint32_t)j, 3u, z);
            for (int k = 0; k < 4 && 4 * j + k < T; ++k) {
                const int t = 4 * j + k;
                const float sd = mean[t] / (snr_v * norm_snr[t]);
                s[v * T + t] += z[k] * sd;
            }
        }
    }
}

// loglinear.fit_wls (loglinear.py:68-105) in closed form.  slope = sum_t a[t] y[t], intercept =
// sum_t b[t] y[t] with a, b fixed by the taus and weights (host, float64); sum a = 0 and sum b = 1,
// so y is taken relative to ln S(tau=0) to keep the float32 sums small.  One lane per voxel, row
// loads through LDS so HBM sees full lines: 4T + 12 bytes per voxel, HBM-bound.
struct WlsCoef {
    float a[QB_MAX_T], b[QB_MAX_T];
    int T, s0;
    float oef_den;  // gamma 4/3 pi dchi hct b0
};

__device__ __forceinline__ float clip_keep_nan(float x, float lo, float hi) {
    return x < lo ? lo : (x > hi ? hi : x);  // np.clip: NaN stays NaN
}

__global__ __launch_bounds__(256) void wls_kernel(WlsCoef c, const float* __restrict__ sig,
                                                  float* __restrict__ out, int64_t N) {
    extern __shared__ float rows[];  // [256][T]
    const int T = c.T;
    const int64_t nblk = (N + 255) / 256;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t v0 = blk * 256;
        const int n = (int)min((int64_t)256, N - v0);
        for (int i = threadIdx.x; i < n * T; i += 256) rows[i] = sig[v0 * T + i];
        __syncthreads();
        if ((int)threadIdx.x < n) {
            const float* r = rows + threadIdx.x * T;
            auto lnS = [](float s) {
                const float y = logf(s);
                return (isnan(y) || isinf(y)) ? 0.0f : y;  // loglinear.py:70-71
            };
            const float y0 = lnS(r[c.s0]);
            float slope = 0.0f, icpt = 0.0f;
            for (int t = 0; t < T; ++t) {
                if (c.a[t] == 0.0f && c.b[t] == 0.0f) continue;  // taus outside the fit
                const float d = lnS(r[t]) - y0;
                slope = fmaf(c.a[t], d, slope);
                icpt = fmaf(c.b[t], d, icpt);
            }
            const float r2p = -slope;
            const float dbv = icpt;  // c - ln S(0)
            const float oef = r2p / (dbv * c.oef_den);
            const int64_t v = v0 + threadIdx.x;
            out[v * 3 + 0] = clip_keep_nan(oef, 0.01f, 0.8f);
            out[v * 3 + 1] = clip_keep_nan(dbv, 0.002f, 0.25f);
            out[v * 3 + 2] = clip_keep_nan(r2p, 1e-2f, 100.0f);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int qbold_wls_fit(const qbold_ctx* ctx, const float* signals, double tau_min, float* out,
                             int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && signals && out, "qbold_wls_fit: bad argument");
    const int T = ctx->dev.T;
    WlsCoef c{};
    c.T = T;
    c.s0 = -1;
    // np.around(np.arange(..., dtype=float32), decimals=7) (loglinear.py:126-127): float32 arithmetic
    double x[QB_MAX_T], w[QB_MAX_T];
    int used = 0;
    double sw = 0, swx = 0;
    const float tmin = (float)tau_min;  // float32 array > python float compares in float32
    for (int t = 0; t < T; ++t) {
        const float r = rintf(ctx->dev.taus[t] * 1e7f) / 1e7f;
        x[t] = (double)r;
        if (r == 0.0f && c.s0 < 0) c.s0 = t;
        w[t] = 0.0;
        if (r > tmin) {
            w[t] = (double)(1.0f / r);  // w = 1 / taus in float32 (loglinear.py:79)
            sw += w[t];
            swx += w[t] * x[t];
            ++used;
        }
    }
    QB_REQUIRE(c.s0 >= 0, "qbold_wls_fit: no tau equals 0 (loglinear.py:92 needs the spin-echo image)");
    QB_REQUIRE(used >= 2, "qbold_wls_fit: fewer than two taus above tau_min");
    const double xm = swx / sw;
    double sxx = 0;
    for (int t = 0; t < T; ++t) sxx += w[t] * (x[t] - xm) * (x[t] - xm);
    for (int t = 0; t < T; ++t) {
        const double a = w[t] * (x[t] - xm) / sxx;  // d slope / d y[t]
        c.a[t] = (float)a;
        c.b[t] = (float)(w[t] / sw - xm * a);       // d intercept / d y[t]
    }
    const qbold_consts& k = ctx->consts;
    c.oef_den = (float)(k.gamma * (4.0 / 3.0) * M_PI * k.dchi * k.hct * k.b0);
    hipLaunchKernelGGL(wls_kernel, dim3(ew_grid(ctx, N, 256)), dim3(256), sizeof(float) * 256 * T,
                       (hipStream_t)stream, c, signals, out, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_normalise(const qbold_ctx* ctx, const float* x, float* out, int64_t N,
                               void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && x && out, "qbold_normalise: bad argument");
    hipLaunchKernelGGL(normalise_kernel, dim3(ew_grid(ctx, N, 256)), dim3(256), 0, (hipStream_t)stream,
                       ctx->dev, x, out, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_transform(const qbold_ctx* ctx, int op, const float* in, float* out, int64_t n,
                               void* stream) {
    QB_NEED_DEVICE(ctx);
    if (n == 0) return QBOLD_OK;
    QB_REQUIRE(n > 0 && in && out, "qbold_transform: bad argument");
    QB_REQUIRE(op >= QBOLD_TRANSFORM_STD && op <= QBOLD_EXP,
               "qbold_transform: unknown op");
    QB_REQUIRE(op < QBOLD_FORWARD_TRANSFORM || op == QBOLD_EXP || (n % 2) == 0,
               "qbold_transform: pair transforms need an even element count");
    hipLaunchKernelGGL(transform_kernel, dim3(ew_grid(ctx, n, 256)), dim3(256), 0, (hipStream_t)stream,
                       op, in, out, n);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_nll_fwd(const qbold_ctx* ctx, const float* x, const float* mask, const float* pred,
                             const float* sigma, float* nll, int64_t N, int S, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && S >= 1 && x && pred && sigma && nll, "qbold_nll_fwd: bad argument");
    const int64_t rows = N * S;
    hipLaunchKernelGGL(nll_kernel, dim3(ew_grid(ctx, rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       ctx->dev, x, mask, pred, sigma, nll, N, rows);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_kl_fwd(const qbold_ctx* ctx, const float* q, const float* prior, const float* zk,
                            int K, uint64_t seed, int64_t voxel0, float* kl, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && K >= 1 && q && prior && kl, "qbold_kl_fwd: bad argument");
    hipLaunchKernelGGL(kl_samples_kernel, dim3(ew_grid(ctx, N, 128)), dim3(128), 0, (hipStream_t)stream,
                       q, prior, zk, K, seed, voxel0, kl, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_kl_closed(const qbold_ctx* ctx, const float* q, const float* prior, float* kl,
                               int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && q && prior && kl, "qbold_kl_closed: bad argument");
    hipLaunchKernelGGL(kl_closed_kernel, dim3(ew_grid(ctx, N, 256)), dim3(256), 0, (hipStream_t)stream,
                       q, prior, kl, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_normals(const qbold_ctx* ctx, uint64_t seed, uint32_t stream_id, int64_t voxel0,
                             int n, float* z, int64_t N, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (N == 0) return QBOLD_OK;
    QB_REQUIRE(N > 0 && n >= 1 && z, "qbold_normals: bad argument");
    hipLaunchKernelGGL(normals_kernel, dim3(ew_grid(ctx, N * ((n + 1) / 2), 256)), dim3(256), 0,
                       (hipStream_t)stream, seed, stream_id, voxel0, n, z, N);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int64_t qbold_noise_workspace_bytes(const qbold_ctx* ctx) {
    if (!ctx) return QBOLD_ERR_INVALID;
    return (int64_t)sizeof(double) * QB_MAX_T * 256 + (int64_t)sizeof(float) * QB_MAX_T;
}

extern "C" int qbold_signal_add_noise(const qbold_ctx* ctx, float* signal, const float* norm_snr_host,
                                      float snr_lo, float snr_hi, uint64_t seed, int64_t voxel0,
                                      void* workspace, int64_t V, void* stream) {
    QB_NEED_DEVICE(ctx);
    if (V == 0) return QBOLD_OK;
    QB_REQUIRE(V > 0 && signal && norm_snr_host && workspace, "qbold_signal_add_noise: bad argument");
    const int T = ctx->dev.T;
    hipStream_t s = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(workspace);
    float* d_snr = reinterpret_cast<float*>(partial + QB_MAX_T * 256);
    QB_HIP(hipMemcpyAsync(d_snr, norm_snr_host, sizeof(float) * T, hipMemcpyHostToDevice, s));
    const int64_t nb = (V + 255) / 256;
    const int nblocks = (int)(nb < 256 ? nb : 256);
    hipLaunchKernelGGL(colsum_kernel, dim3(nblocks), dim3(256), sizeof(double) * 256, s, signal, T, V,
                       partial);
    QB_HIP(hipGetLastError());
    hipLaunchKernelGGL(add_noise_kernel, dim3(ew_grid(ctx, V, 256)), dim3(256), 0, s, signal, T, V,
                       partial, nblocks, d_snr, snr_lo, snr_hi, seed, voxel0);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

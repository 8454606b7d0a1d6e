// signal_kernels.hip -- the batched qBOLD forward signal model and its vector-Jacobian product.
//
// Replaces SignalGenerationLayer.call (signals.py:55-114,137-138) for noise-free, aligned inputs:
// one lane per (OEF, DBV) pair, T signals per lane.  The reference materialises [V,T,129]
// float32 intermediates (signals.py:168-171); here the integral is a 4 KiB LDS table (or, in
// literal mode, a register-resident 129-node Simpson sum), inputs are one coalesced 8-byte load
// per lane and the [V][T] rows leave through an LDS transpose as full 16-byte stores.
//
// Roofline: table mode moves 8 + 4T bytes per voxel for ~25 VALU ops per (voxel, tau): at T=11,
// 52 B and ~300 ops per voxel -> ~6 flop/B, under the f32 machine balance (~20 flop/B), so the
// kernel is HBM-bound.  Literal mode is VALU-bound (129 Bessel evaluations per (voxel, tau)).
#include "qbold_ctx.h"

namespace {

constexpr int kBlock = 256;

// EX: the options optimal.yaml disables (signals.py:64-96) -- per-voxel haematocrit and the
// misalignment augmentation (images t > from_idx[v] come from alt[v]); any of the three may be null.
template <bool LITERAL, bool EX>
__global__ __launch_bounds__(kBlock) void signal_fwd_kernel(QbDev c, const float4* __restrict__ g_tab,
                                                            const float2* __restrict__ oef_dbv,
                                                            float* __restrict__ signal, int64_t V,
                                                            int vec_ok, const float* __restrict__ hct,
                                                            const float2* __restrict__ alt,
                                                            const int* __restrict__ from_idx) {
    extern __shared__ __align__(16) unsigned char smem[];
    qb::FwdLds* L = reinterpret_cast<qb::FwdLds*>(smem);
    float* stage = reinterpret_cast<float*>(smem + sizeof(qb::FwdLds));  // [kBlock][T]
    qb::fwd_lds_fill(L, g_tab, true);  // literal tables also serve |x| beyond the table
    __syncthreads();

    const int T = c.T;
    const int64_t nblk = (V + kBlock - 1) / kBlock;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t v0 = blk * kBlock;
        const int64_t v = v0 + threadIdx.x;
        if (v < V) {
            const float2 p = oef_dbv[v];
            if constexpr (!EX) {
                const qb::FwdVox fv = qb::fwd_vox(c, p.x, p.y);
                for (int t = 0; t < T; ++t)
                    stage[threadIdx.x * T + t] = qb::fwd_signal<LITERAL>(L, c, fv, t);
            } else {
                const int split = (alt && from_idx) ? from_idx[v] : T;
                const float2 p2 = split < T - 1 ? alt[v] : p;
                const qb::FwdVox fa = hct ? qb::fwd_vox_hct(c, p.x, p.y, hct[v]) : qb::fwd_vox(c, p.x, p.y);
                const qb::FwdVox fb = hct ? qb::fwd_vox_hct(c, p2.x, p2.y, hct[v]) : qb::fwd_vox(c, p2.x, p2.y);
                for (int t = 0; t < T; ++t)
                    stage[threadIdx.x * T + t] = qb::fwd_signal<LITERAL>(L, c, t > split ? fb : fa, t);
            }
        }
        __syncthreads();
        // rows v0 .. v0+n-1 are contiguous in global memory: n*T floats from signal + v0*T
        const int n = (int)min((int64_t)kBlock, V - v0);
        const int total = n * T;
        float* dst = signal + v0 * T;  // v0*T*4 bytes is a multiple of 16 (kBlock*4 is)
        const int nvec = vec_ok ? total >> 2 : 0;  // 16-byte stores need a 16-byte aligned base
        for (int i = threadIdx.x; i < nvec; i += kBlock)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(stage)[i];
        for (int i = (nvec << 2) + threadIdx.x; i < total; i += kBlock) dst[i] = stage[i];
        __syncthreads();
    }
}

// grad_oef_dbv[v] = sum_t grad_signal[v][t] * d signal[v][t] / d (oef, dbv)
template <bool LITERAL>
__global__ __launch_bounds__(kBlock) void signal_bwd_kernel(QbDev c, const float4* __restrict__ g_tab,
                                                            const float2* __restrict__ oef_dbv,
                                                            const float* __restrict__ grad_signal,
                                                            float2* __restrict__ grad_in, int64_t V,
                                                            int vec_ok) {
    extern __shared__ __align__(16) unsigned char smem[];
    qb::FwdLds* L = reinterpret_cast<qb::FwdLds*>(smem);
    float* stage = reinterpret_cast<float*>(smem + sizeof(qb::FwdLds));  // [kBlock][T]
    qb::fwd_lds_fill(L, g_tab, true);
    __syncthreads();

    const int T = c.T;
    const int64_t nblk = (V + kBlock - 1) / kBlock;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t v0 = blk * kBlock;
        const int n = (int)min((int64_t)kBlock, V - v0);
        const int total = n * T;
        const float* src = grad_signal + v0 * T;
        const int nvec = vec_ok ? total >> 2 : 0;
        for (int i = threadIdx.x; i < nvec; i += kBlock)
            reinterpret_cast<float4*>(stage)[i] = reinterpret_cast<const float4*>(src)[i];
        for (int i = (nvec << 2) + threadIdx.x; i < total; i += kBlock) stage[i] = src[i];
        __syncthreads();
        const int64_t v = v0 + threadIdx.x;
        if (v < V) {
            const float2 p = oef_dbv[v];
            const float oef = p.x, dbv = p.y;
            const qb::FwdVox fv = qb::fwd_vox(c, oef, dbv);
            const float dbw = c.include_blood ? c.m_bld_nb : 1.0f;
            const float dg = (c.include_blood && oef != 0.0f) ? 2.0f * fv.g / oef : 0.0f;
            float g_oef = 0.0f, g_dbv = 0.0f;
            for (int t = 0; t < T; ++t) {
                const float tau = c.taus[t];
                float tissue, dt_doef, dt_ddbv;
                if (c.full_model) {
                    float dF;
                    float F = qb::tissue_F<LITERAL, true>(L, c, tau * fv.dw, &dF);
                    tissue = __expf(-dbv * F) * c.e_te_r2t;
                    dt_ddbv = -F * tissue;
                    dt_doef = -dbv * dF * (tau * c.dw_coef) * tissue;
                } else {
                    float tc = 1.0f / fv.dw;
                    float k = tau * fv.dw;
                    float e = __expf(c.r2t_te);
                    if (fabsf(tau) < tc) {
                        tissue = e * __expf(-(0.3f * (k * k)) * dbv);
                        dt_ddbv = -0.3f * k * k * tissue;
                        dt_doef = -0.6f * k * (tau * c.dw_coef) * dbv * tissue;
                    } else {
                        tissue = e * __expf(dbv - k * dbv);
                        dt_ddbv = (1.0f - k) * tissue;
                        dt_doef = -(c.dw_coef * dbv * tau) * tissue;
                    }
                }
                float blood = 0.0f, db_doef = 0.0f;
                if (c.include_blood) {
                    blood = c.e_r2b_te * __expf(-fv.g * c.blood_B[t]);
                    db_doef = -dg * c.blood_B[t] * blood;
                }
                const float gs = stage[threadIdx.x * T + t];
                g_oef += gs * (fv.tw * dt_doef + fv.bw * db_doef);
                g_dbv += gs * (fv.tw * dt_ddbv + dbw * (blood - tissue));
            }
            grad_in[v] = make_float2(g_oef, g_dbv);
        }
        __syncthreads();
    }
}

int grid_for(const qbold_ctx* ctx, int64_t V, int per_cu) {
    int64_t nblk = (V + kBlock - 1) / kBlock;
    int64_t cap = (int64_t)ctx->num_cus * per_cu;
    return (int)(nblk < cap ? (nblk > 0 ? nblk : 1) : cap);
}

}  // namespace

extern "C" int qbold_signal_fwd(const qbold_ctx* ctx, const float* oef_dbv, float* signal,
                                int64_t V, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(V >= 0, "qbold_signal_fwd: negative V");
    if (V == 0) return QBOLD_OK;
    QB_REQUIRE(oef_dbv && signal, "qbold_signal_fwd: null buffer");
    const size_t smem = sizeof(qb::FwdLds) + sizeof(float) * kBlock * ctx->dev.T;
    hipStream_t s = (hipStream_t)stream;
    const float2* in = reinterpret_cast<const float2*>(oef_dbv);
    QB_REQUIRE(reinterpret_cast<uintptr_t>(oef_dbv) % 8 == 0, "qbold_signal_fwd: oef_dbv must be 8-byte aligned");
    const int vec_ok = reinterpret_cast<uintptr_t>(signal) % 16 == 0;
    if (ctx->dev.tissue_mode == QBOLD_TISSUE_LITERAL)
        hipLaunchKernelGGL((signal_fwd_kernel<true, false>), dim3(grid_for(ctx, V, 8)), dim3(kBlock), smem, s,
                           ctx->dev, ctx->d_tab, in, signal, V, vec_ok, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((signal_fwd_kernel<false, false>), dim3(grid_for(ctx, V, 8)), dim3(kBlock), smem, s,
                           ctx->dev, ctx->d_tab, in, signal, V, vec_ok, nullptr, nullptr, nullptr);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_signal_fwd_ex(const qbold_ctx* ctx, const float* oef_dbv, const float* hct,
                                   const float* alt_oef_dbv, const int32_t* from_index, float* signal,
                                   int64_t V, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(V >= 0, "qbold_signal_fwd_ex: negative V");
    if (V == 0) return QBOLD_OK;
    QB_REQUIRE(oef_dbv && signal, "qbold_signal_fwd_ex: null buffer");
    QB_REQUIRE((alt_oef_dbv == nullptr) == (from_index == nullptr),
               "qbold_signal_fwd_ex: alt_oef_dbv and from_index come together");
    QB_REQUIRE(reinterpret_cast<uintptr_t>(oef_dbv) % 8 == 0 && reinterpret_cast<uintptr_t>(alt_oef_dbv) % 8 == 0,
               "qbold_signal_fwd_ex: (OEF, DBV) buffers must be 8-byte aligned");
    const size_t smem = sizeof(qb::FwdLds) + sizeof(float) * kBlock * ctx->dev.T;
    hipStream_t s = (hipStream_t)stream;
    const float2* in = reinterpret_cast<const float2*>(oef_dbv);
    const float2* alt = reinterpret_cast<const float2*>(alt_oef_dbv);
    const int vec_ok = reinterpret_cast<uintptr_t>(signal) % 16 == 0;
    if (ctx->dev.tissue_mode == QBOLD_TISSUE_LITERAL)
        hipLaunchKernelGGL((signal_fwd_kernel<true, true>), dim3(grid_for(ctx, V, 8)), dim3(kBlock), smem, s,
                           ctx->dev, ctx->d_tab, in, signal, V, vec_ok, hct, alt, from_index);
    else
        hipLaunchKernelGGL((signal_fwd_kernel<false, true>), dim3(grid_for(ctx, V, 8)), dim3(kBlock), smem, s,
                           ctx->dev, ctx->d_tab, in, signal, V, vec_ok, hct, alt, from_index);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

extern "C" int qbold_signal_bwd(const qbold_ctx* ctx, const float* oef_dbv, const float* grad_signal,
                                float* grad_oef_dbv, int64_t V, void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_REQUIRE(V >= 0, "qbold_signal_bwd: negative V");
    if (V == 0) return QBOLD_OK;
    QB_REQUIRE(oef_dbv && grad_signal && grad_oef_dbv, "qbold_signal_bwd: null buffer");
    const size_t smem = sizeof(qb::FwdLds) + sizeof(float) * kBlock * ctx->dev.T;
    hipStream_t s = (hipStream_t)stream;
    const float2* in = reinterpret_cast<const float2*>(oef_dbv);
    float2* gout = reinterpret_cast<float2*>(grad_oef_dbv);
    QB_REQUIRE(reinterpret_cast<uintptr_t>(oef_dbv) % 8 == 0 && reinterpret_cast<uintptr_t>(grad_oef_dbv) % 8 == 0,
               "qbold_signal_bwd: (OEF, DBV) buffers must be 8-byte aligned");
    const int vec_ok = reinterpret_cast<uintptr_t>(grad_signal) % 16 == 0;
    if (ctx->dev.tissue_mode == QBOLD_TISSUE_LITERAL)
        hipLaunchKernelGGL(signal_bwd_kernel<true>, dim3(grid_for(ctx, V, 8)), dim3(kBlock), smem, s,
                           ctx->dev, ctx->d_tab, in, grad_signal, gout, V, vec_ok);
    else
        hipLaunchKernelGGL(signal_bwd_kernel<false>, dim3(grid_for(ctx, V, 8)), dim3(kBlock), smem, s,
                           ctx->dev, ctx->d_tab, in, grad_signal, gout, V, vec_ok);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

// vi_kernels.hip -- the whole voxel-ELBO hot path behind one entry point (qbold_vi_fwd).
//
// Narrow encoders (U <= 64, T = 11 or 24: BASELINE configs 1, 2, 4, 5) -- one launch, vi_fwd_kernel.  Per
// 16-voxel wave tile: load x -> normalise -> encoder stream 2 on the f16 matrix cores (encoder_core.h: operands
// split into two f16 halves, three v_mfma_f32_16x16x32_f16 passes hi.hi + hi.lo + lo.hi with f32 accumulation,
// float32-grade products; QBOLD_ENC_BF16: operands rounded to bf16, one pass) -> posterior parameters and log
// sigma stay in registers -> S reparameterised draws through the forward model + K-draw Monte-Carlo KL on the
// vector pipe (elbo_core.h) -> per-voxel (nll, kl), posterior parameters and the three masked sums.  This is the
// reference's full_model([data, mask]) + fine_tune_loss_fn + kl_loss (model.py:239-286, 527-568, 654-665)
// evaluated for (N,1,1,1,T) voxel batches, with ELBO = nll + kl as train.py:351.
//
// HBM traffic per voxel (T = 11): read x 44 + mask 4 + prior 20, write q 20 + (nll, kl) 8 = 96 B; the weight
// image (~145 KB of split-f16 fragments), the F(x) table (4 KB) and constants are LDS / SGPR resident.  Work per
// voxel at S = 32, K = 70: 60.8 kFLOP of encoder products (x 3 passes on the matrix pipe) and ~30 kFLOP of
// sampling: compute-bound by the vector pipe's issue rate (MEASUREMENTS.md 4.4), ~900 flop/B against a machine balance
// of ~20 flop/B.
//
// One 1024-thread workgroup per CU (the weight image takes ~145 KB of the 160 KB LDS), 128 VGPRs, so four waves
// share each SIMD: while some are in their sampling phase others own the SIMD's matrix pipe, and the LDS-table /
// transcendental latencies of the sampling phase are covered by thread-level parallelism.
//
// An activation or weight outside the f16 operand range (|v| > 65504) cannot be split: the pack kernel records
// it in the image's flag slot, the encoder tracks the largest activation it split, and the voxel's terms come
// out NaN -- never a silent clamp (include/qbold_hip.h, operand range; ops.vi_fwd(range_check=True) falls back to
// the exact-f32 layer-wise path).
//
// Wide encoders (U = 256, T <= 16 or 49..64: BASELINE config 3) -- two launches: the one-launch encoder of
// wide_fused_kernels.hip writes q and log sigma into the caller's workspace (qbold_vi_workspace_bytes), the
// compile-time-T ELBO kernel of elbo_kernels.hip reads them back.
#include "elbo_core.h"
// Wave priorities (s_setprio): a wave starts a tile (signal loads, normalisation) at 3, runs its encoder
// phase at 2 (3 inside the MFMA chains), its likelihood draws at 1, its KL draws at 3 and the tile's tail
// at 0 (elbo_core.h).  The four waves of a SIMD are in different phases most of the time; preferring the
// one that feeds the matrix pipe, and the one about to finish, keeps the matrix pipe busy while the
// others fill the VALU: 0.610 -> 0.572 ms per 1 M voxels.
#ifndef QB_PRIO_TILE_START
#define QB_PRIO_TILE_START 3
#endif
#define QB_ENC_BASE_PRIO 2
#include "encoder_core.h"
#include "qbold_ctx.h"

namespace qb {
int check_encoder_shape(const qbold_ctx* ctx, const qbold_encoder_shape* s);
bool elbo_fast_path(const qbold_ctx* ctx);
bool elbo_logsigma_path(const qbold_ctx* ctx);
int elbo_fwd_launch(const qbold_ctx* ctx, const float* x, const float* mask, const float* q, const float* prior,
                    const float* sigma, bool sigma_is_log, const float* zs, const float* zk, int S, int K,
                    uint64_t seed, int64_t voxel0, float* nll_kl, double* sums, void* workspace, int64_t N,
                    hipStream_t s);
bool wide_fused_supported(const qbold_encoder_shape* s);
int wide_fused_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed, const float* x,
                   float* out_q, float* out_log_sigma, int64_t N, hipStream_t s);
}

namespace {

using qb::EncLayout;
using qb::f32x4;

constexpr int kBlock = 1024;
// threads per workgroup of the 24-tau instantiations: with 1,024 (128 VGPRs) that protocol's 34 data registers leave the
// kernel ~35 registers short around its sampling phase (140 bytes of scratch per lane, re-read every tile: 4.5 x the
// algorithmic HBM bytes); 768 threads get 168 VGPRs and three waves per SIMD
#ifndef QB_VI_BLOCK_24
#define QB_VI_BLOCK_24 768
#endif

template <int T, int SE, bool GT>
struct ViLds { using type = qb::FwdLds; };
template <int T, int SE>
struct ViLds<T, SE, true> { using type = qb::GtLds<T, SE>; };
constexpr size_t kLdsLimit = 160 * 1024;   // gfx950: LDS per workgroup

// GT: the sampling fast path reads the per-tau OEF-indexed table (GtLds) instead of the x-indexed one (FwdLds);
// requires FAST and a compile-time spin-echo index.
// MIR: the protocol mirrors about the spin echo (qbold_ctx::grid_mirrors): mirrored tau pairs are evaluated once and
// scored as one merged data point (elbo_core.h, prepare_lik); GT implies it.
template <int T, int NL, int SE, bool FAST, bool LITERAL, bool BF, bool GT = false, bool MIR = false, int BLK = kBlock>
__global__ __launch_bounds__(BLK) void vi_fwd_kernel(
    QbDev c, const float4* __restrict__ g_tab, const float* __restrict__ packed,
    const float* __restrict__ x, const float* __restrict__ mask, const float* __restrict__ prior,
    int S, int K, uint64_t seed, int64_t voxel0, float* __restrict__ q_out,
    float2* __restrict__ nll_kl, double* __restrict__ partials, int64_t N) {
    // compile-time weight-image layout: every LDS offset below folds into an instruction immediate
    constexpr EncLayout e = qb::make_enc_layout(T, 64, NL);
    // the sampling phase's table: per-tau OEF-indexed rows (GtLds) on the fast path with a compile-time spin-echo
    // index, the x-indexed table + literal nodes (FwdLds) otherwise; g_tab points at the matching device table
    static_assert(!GT || (FAST && SE >= 0 && qb::gtab_segs(T) > 0), "GT needs the fast path with a compile-time spin echo");
    static_assert(!MIR || (FAST && SE >= 0), "merged mirror pairs: fast path with a compile-time spin echo");
    constexpr bool kMir = GT || MIR;
    using Lds = typename ViLds<T, SE, GT>::type;
    extern __shared__ __align__(16) unsigned char smem[];
    float* lds_w = reinterpret_cast<float*>(smem);
    Lds* L = reinterpret_cast<Lds*>(smem + sizeof(float) * e.total);
    double* red = reinterpret_cast<double*>(smem + sizeof(float) * e.total + sizeof(Lds));
    constexpr int kWaves = BLK / 64;
    qb::copy_to_lds<BLK>(lds_w, packed, e.total / 4);
    if constexpr (qb::IsGtLds<Lds>::value) {
        qb::gt_lds_fill(L, g_tab, c);
    } else {
        qb::fwd_lds_fill(L, g_tab, true);
        if (threadIdx.x < QB_MAX_T) L->blood_B[threadIdx.x] = c.blood_B[threadIdx.x];
    }
    __syncthreads();

    constexpr int HT = (5 + T + 15) / 16;
    const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s_nll = 0.0f, s_kl = 0.0f, s_m = 0.0f;
    const int64_t ntile = (N + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < ntile;
         tile += (int64_t)gridDim.x * kWaves) {
        // The lane index is made opaque once per tile: everything derived from it (the encoder's LDS fragment
        // addresses, the voxel's row pointers) is then recomputed per tile -- a handful of vector instructions --
        // instead of being hoisted out of the tile loop as 20-70 loop-invariant registers that the sampling phase
        // (which needs none of them) has to spill and reload around itself.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int g = lane >> 4, i = lane & 15;
        const int64_t v = tile * 16 + i;
        const int64_t vc = v < N ? v : N - 1;
        float o[5 + T];
        // largest activation this lane split (operand range guard, encoder_core.h); starts from the image's flag
        // slot: 0, or the largest weight the pack kernel could not split
        float amax = lds_w[e.flag];
        {
            __builtin_amdgcn_s_setprio(QB_PRIO_TILE_START);
            float xv[T], nv[T];
#pragma unroll
            for (int t = 0; t < T; ++t) xv[t] = x[vc * T + t];
            qb::normalise<T, SE>(c, xv, nv);
            __builtin_amdgcn_s_setprio(2);
            f32x4 b[4];
            qb::dense_first<T, BF>(lds_w + e.first_A, lds_w + e.first_b, nv, b, lane);
            if (!QB_ABLATE(c, 1) && qb_phase_fence())
#pragma unroll
                for (int l = 0; l < NL; ++l) qb::block_stream2<BF>(lds_w + e.blk0 + l * e.blk_stride, b, lane, &amax);
            f32x4 hd[HT];
            qb::dense_head<HT, BF>(lds_w + e.head_A, lds_w + e.head_b, b, hd, lane, &amax);
            qb::gather_head<5 + T, HT>(hd, o);
            __builtin_amdgcn_s_setprio(0);
        }
        if (v < N && !QB_ABLATE(c, 2) && qb_phase_fence()) {
            // x is read again (an L1/L2 hit) rather than held in 11 VGPRs across the encoder; the
            // empty asm keeps the compiler from merging the two reads
            const float* xr = x + v * T;
            asm volatile("" : "+v"(xr));
            float xv[T], sv[T], qv[5];
#pragma unroll
            for (int t = 0; t < T; ++t) xv[t] = xr[t];
#pragma unroll
            for (int k = 0; k < 5; ++k) qv[k] = o[k];
#pragma unroll
            for (int t = 0; t < T; ++t) sv[t] = o[5 + t];  // log sigma; sigma = exp(.), model.py:214
            // the mask: the likelihood needs it only on log data (model.py:548); the sums read it again after the draws
            // (a second load through an opaque pointer) instead of carrying it through the loops
            qb::VoxelLik<T> lik;
            qb::prepare_lik<T, SE, true, (FAST && SE >= 0), FAST, kMir>(c, xv, sv, FAST ? 1.0f : (mask ? mask[v] : 1.0f), lik);
            const qb::LogitMvn qm = qb::make_mvn(qv);
            // posterior parameters leave before the draws (lane group 1), so that they do not live through them
            if (g == 1 && q_out) {
#pragma unroll
                for (int k = 0; k < 5; ++k) q_out[v * 5 + k] = qv[k];
            }
            // an activation beyond the f16 operand range: this voxel's terms become NaN (never a silent clamp) -- through
            // the per-draw constant, which every draw's NLL adds
            if (!BF && qb::split_overflowed(amax)) lik.log_s_sum = __builtin_nanf("");
            float nll_part, kl_part;
            qb::voxel_mc_sums<T, SE, FAST, LITERAL, kMir>(L, c, lik, qm, prior + v * 5, S, K, nullptr, nullptr, seed,
                                                    (uint64_t)(voxel0 + v), g, nll_part, kl_part);
            const float nll = qb::voxel_sum(nll_part) / (float)S;
            const float kl = K > 0 ? qb::voxel_sum(kl_part) / (float)K : 0.0f;
            if (g == 0) {
                const float* mp = mask;
                asm volatile("" : "+v"(mp));
                const float m = mp ? mp[v] : 1.0f;
                if (nll_kl) nll_kl[v] = make_float2(nll, kl);
                s_nll += nll * m;              // model.py:564
                s_kl += m > 0.0f ? kl : 0.0f;  // model.py:661
                s_m += m;
            }
        }
    }
    qb::block_partials(red, s_nll, s_kl, s_m, partials);
}

}  // namespace

namespace {
// Wide encoders (BASELINE config 3): the path is two launches -- the one-launch encoder of
// wide_fused_kernels.hip, then the ELBO kernel reading the heads (q, log sigma) back from the workspace.
// Workspace: [per-workgroup partials][log sigma N T floats][q N 5 floats (used when q_out is NULL)].
inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }
bool wide_vi_path(const qbold_ctx* ctx, const qbold_encoder_shape* shape) {
    return qb::wide_fused_supported(shape) && ctx && shape->T == ctx->dev.T && qb::elbo_logsigma_path(ctx);
}
}  // namespace

extern "C" int64_t qbold_vi_workspace_bytes(const qbold_ctx* ctx, const qbold_encoder_shape* shape, int64_t N) {
    if (!ctx || !shape || N < 0) return QBOLD_ERR_INVALID;
    const int64_t part = align256(qbold_elbo_workspace_bytes(ctx));
    if (!wide_vi_path(ctx, shape)) return part;
    return part + align256(N * (int64_t)shape->T * 4) + align256(N * 5 * 4);
}

extern "C" int qbold_vi_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                            const float* packed, const float* x, const float* mask,
                            const float* prior, int S, int K, uint64_t seed, int64_t voxel0,
                            float* q_out, float* nll_kl, double* sums, void* workspace, int64_t N,
                            void* stream) {
    QB_NEED_DEVICE(ctx);
    QB_RELU_ONLY(shape, "qbold_vi_fwd");
    if (wide_vi_path(ctx, shape)) {
        QB_REQUIRE(N >= 0 && S >= 1 && K >= 0, "qbold_vi_fwd: need N >= 0, S >= 1, K >= 0");
        QB_REQUIRE(sums && workspace, "qbold_vi_fwd: null sums/workspace");
        QB_REQUIRE(N == 0 || (packed && x && prior), "qbold_vi_fwd: null input buffer");
        QB_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 256 == 0 && reinterpret_cast<uintptr_t>(packed) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(x) % 16 == 0,
                   "qbold_vi_fwd: wide shapes need a 256-byte aligned workspace of qbold_vi_workspace_bytes() and "
                   "16-byte aligned packed / x");
        char* w = reinterpret_cast<char*>(workspace);
        float* ls = reinterpret_cast<float*>(w + align256(qbold_elbo_workspace_bytes(ctx)));
        float* qbuf = q_out ? q_out : reinterpret_cast<float*>(reinterpret_cast<char*>(ls) + align256(N * (int64_t)shape->T * 4));
        if (N > 0) {
            const int rc = qb::wide_fused_fwd(ctx, shape, packed, x, qbuf, ls, N, (hipStream_t)stream);
            if (rc) return rc;
        }
        return qb::elbo_fwd_launch(ctx, x, mask, qbuf, prior, ls, true, nullptr, nullptr, S, K, seed, voxel0, nll_kl,
                                   sums, workspace, N, (hipStream_t)stream);
    }
    int rc = qb::check_encoder_shape(ctx, shape);
    if (rc) return rc;
    QB_REQUIRE(N >= 0 && S >= 1 && K >= 0, "qbold_vi_fwd: need N >= 0, S >= 1, K >= 0");
    QB_REQUIRE(sums && workspace, "qbold_vi_fwd: null sums/workspace");
    QB_REQUIRE(N == 0 || (packed && x && prior), "qbold_vi_fwd: null input buffer");
    hipStream_t s = (hipStream_t)stream;
    double* partials = reinterpret_cast<double*>(workspace);
    const int64_t ntile = (N + 15) / 16;
    const int blk = shape->T == 24 ? QB_VI_BLOCK_24 : kBlock, waves = blk / 64;
    const int64_t nblk = (ntile + waves - 1) / waves;
    const int grid = (int)(nblk < ctx->num_cus ? (nblk > 0 ? nblk : 1) : ctx->num_cus);
    const bool lit = ctx->dev.tissue_mode == QBOLD_TISSUE_LITERAL;
    float2* out = reinterpret_cast<float2*>(nll_kl);
    const bool bf = shape->precision == QBOLD_ENC_BF16;
    // the reduced-precision encoder is built for the table-mode fast path (the bench / training
    // configuration); the literal and generic paths exist to reproduce float32 semantics
    QB_REQUIRE(!bf || qb::elbo_fast_path(ctx),
               "qbold_vi_fwd: QBOLD_ENC_BF16 needs the table-mode Gaussian fast path");
#define QB_LAUNCH_VI(TT, NL, SE, FAST, LIT) QB_LAUNCH_VI_GT(TT, NL, SE, FAST, LIT, false, false)
#define QB_LAUNCH_VI_GT(TT, NL, SE, FAST, LIT, GT, MIR)                                              \
    do {                                                                                          \
        using LdsT = typename ViLds<TT, SE, GT>::type;                                             \
        constexpr size_t smem = sizeof(float) * qb::make_enc_layout(TT, 64, NL).total + sizeof(LdsT) + \
                                sizeof(double) * 3 * (kBlock / 64);                                \
        static_assert(smem <= kLdsLimit, "weight image + sampling table exceed the LDS");          \
        const float4* tab = qb::IsGtLds<LdsT>::value ? ctx->d_gtab : ctx->d_tab;                    \
        constexpr int BLKT = TT == 24 ? QB_VI_BLOCK_24 : kBlock;                                   \
        auto k = vi_fwd_kernel<TT, NL, SE, FAST, LIT, false, GT, MIR, BLKT>;                       \
        if constexpr (FAST) { if (bf) k = vi_fwd_kernel<TT, NL, SE, FAST, LIT, true, GT, MIR, BLKT>; }   \
        QB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k),                              \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));       \
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLKT), smem, s, ctx->dev, tab, packed,              \
                           x, mask, prior, S, K, seed, voxel0, q_out, out, partials, N);          \
    } while (0)
    const bool fast = qb::elbo_fast_path(ctx);
    // SEC: the protocol's spin-echo index (tau = 0), folded at compile time when the context agrees
#define QB_DISPATCH_VI(TT, NL, SEC)                                               \
    do {                                                                          \
        if (qb::gtab_segs(TT) > 0 && fast && ctx->dev.se_idx == SEC && !ctx->dev.multi_norm && !(ctx->kernel_sel & 4) && ctx->gtab_ok && !(ctx->kernel_sel & 8)) QB_LAUNCH_VI_GT(TT, NL, SEC, true, false, (qb::gtab_segs(TT) > 0), false);   \
        else if (fast && ctx->dev.se_idx == SEC && !ctx->dev.multi_norm && !(ctx->kernel_sel & 4) && ctx->grid_mirrors) QB_LAUNCH_VI_GT(TT, NL, SEC, true, false, false, true);   \
        else if (fast && ctx->dev.se_idx == SEC && !ctx->dev.multi_norm && !(ctx->kernel_sel & 4)) QB_LAUNCH_VI(TT, NL, SEC, true, false);   \
        else if (fast) QB_LAUNCH_VI(TT, NL, -1, true, false);                     \
        else if (lit && ctx->dev.se_idx == SEC && !ctx->dev.multi_norm) QB_LAUNCH_VI(TT, NL, SEC, false, true);   \
        else if (lit) QB_LAUNCH_VI(TT, NL, -1, false, true);                      \
        else QB_LAUNCH_VI(TT, NL, -1, false, false);                              \
    } while (0)
#ifdef QB_VI_PROBE   // scripts/dev/resources.sh: compile the optimal.yaml depth only (register / scratch reports in seconds)
    if (shape->T == 11 && shape->L == 2) QB_DISPATCH_VI(11, 2, 2);
    else if (shape->T == 24 && shape->L == 2) QB_DISPATCH_VI(24, 2, 7);
#else
    if (shape->T == 11 && shape->L == 1) QB_DISPATCH_VI(11, 1, 2);
    else if (shape->T == 11 && shape->L == 2) QB_DISPATCH_VI(11, 2, 2);
    else if (shape->T == 24 && shape->L == 1) QB_DISPATCH_VI(24, 1, 7);
    else if (shape->T == 24 && shape->L == 2) QB_DISPATCH_VI(24, 2, 7);
#endif
    else {
        qb::set_error("qbold_vi_fwd: kernels are built for T = 11 or 24 taus, L = 1 or 2");
        return QBOLD_ERR_UNSUPPORTED;
    }
#undef QB_DISPATCH_VI
#undef QB_LAUNCH_VI
#undef QB_LAUNCH_VI_GT
    QB_HIP(hipGetLastError());
    hipLaunchKernelGGL(qb::reduce_partials_kernel, dim3(1), dim3(192), 0, s, partials, grid, sums);
    QB_HIP(hipGetLastError());
    return QBOLD_OK;
}

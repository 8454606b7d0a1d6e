/*
 * qbold_hip.h -- C ABI of libqbold_hip.so: the MI355X (gfx950) implementation of the qBOLD-VI
 * voxel-wise amortised-VI hot path.
 *
 * The reference (wearepal/qBOLD-VI) has no FFI/plugin interface: the path sits behind Python
 * classes used as Keras layers / loss callables.  Each entry point below names the reference
 * interface (file:line, relative to the reference repository root) whose arithmetic it replaces;
 * the Python host mirror in qbold_vi_amd/ keeps the reference's class and method names and calls
 * these functions through ctypes (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - every buffer is a caller-owned DEVICE pointer (float32, row-major, channel-last) unless
 *     a parameter is documented as host; nothing is allocated per call;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     asynchronous on that stream, and the library never synchronises the device;
 *   - return value: 0 on success, a negative qbold_status otherwise; functions never throw;
 *   - a context is bound to one device, immutable after creation and re-entrant.
 *
 * Tensor conventions (as the reference): posterior/prior parameters are
 * [mu_oef, raw_s_oef, mu_dbv, raw_s_dbv, raw_c] (model.py:26-31,380-383); (OEF,DBV) pairs are
 * interleaved [V][2] (signals.py:75-77); signals are [V][T] with T = number of taus.
 */
#ifndef QBOLD_HIP_H
#define QBOLD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QBOLD_MAX_T 64
#define QBOLD_ABI_VERSION 5

typedef enum {
    QBOLD_OK = 0,
    QBOLD_ERR_INVALID = -1,     /* bad argument (null pointer, size, unsupported shape) */
    QBOLD_ERR_HIP = -2,         /* a HIP runtime call failed; see qbold_last_error() */
    QBOLD_ERR_UNSUPPORTED = -3, /* configuration outside what the kernels were built for */
    QBOLD_ERR_NO_DEVICE = -4
} qbold_status;

/* System constants: INI `config` [DEFAULT] (config:1-38) as parsed by
 * SignalGenerationLayer.__init__ (signals.py:18-53). */
typedef struct {
    double gamma, b0, dchi, te, r2t, tr, ti, t1b, hct;
    double tau_start, tau_end, tau_step; /* tf.range(...) grid, signals.py:34-35 */
    int32_t full_model;                  /* signals.py:192: Simpson/Bessel tissue model or log-linear */
    int32_t include_blood;               /* signals.py:100 */
} qbold_consts;

/* Likelihood / normalisation switches of EncoderTrainer.__init__ (model.py:54-95). */
typedef struct {
    int32_t multi_image_normalisation; /* model.py:102-106, 540-545 */
    int32_t predict_log_data;          /* model.py:547-549 */
    int32_t use_student_t;             /* model.py:557: student_t_df is not None and < 50 */
    double student_t_df;
} qbold_loss_cfg;

/* How the tissue integral F(x) of signals.py:159-185 is evaluated. */
typedef enum {
    QBOLD_TISSUE_TABLE = 0,   /* cubic-Hermite table of the reference's float32 Simpson-129 sum */
    QBOLD_TISSUE_LITERAL = 1  /* the 129-node Simpson sum with Cephes j0f, per (voxel, tau) */
} qbold_tissue_mode;

/* Arithmetic of the fused voxel-wise encoder kernels (qbold_encoder_fwd, qbold_vi_fwd, qbold_encoder_wide_fwd,
 * qbold_encoder_fused_fwd).
 *
 * QBOLD_ENC_F32 operand range.  Every weight and activation x enters the matrix cores as hi = f16(x) plus a lo half
 * carrying the next 11 bits; products hi.hi + hi.lo + lo.hi accumulate in float32 (the lo.lo term, <= 2^-22
 * relative, is dropped).  Consequences the float32 reference does not share:
 *   - |x| > 65504 (the largest f16): hi overflows to inf.  The heads then come out non-finite and, because
 *     nll * mask is NaN for a NaN nll whatever the mask, so do the three sums: NON-FINITE SUMS ARE THE STATUS
 *     CHANNEL for this condition (the library never clamps silently).  The exact-float32 layer-wise path
 *     (qbold_encoder_train_fwd + qbold_elbo_fwd) has no such limit; the Python mirror's
 *     Context.vi_fwd(range_check=True) / FineTuner.elbo fall back to it.
 *   - 6.1e-5 <= |x| <= 65504: 22 significant bits (relative error <= ~7e-7 per product).
 *   - |x| < 6.1e-5: both halves are f16 subnormals, absolute resolution 2^-35 = 2.9e-11 (LDS-resident kernels,
 *     lo scaled by 2^11) or 2^-25 = 3e-8 for activations in qbold_encoder_fused_fwd (unscaled lo; its weights
 *     are pre-scaled per dense op by a power of two, so their lo halves stay normal) -- absolute errors far
 *     below float32 rounding of an O(1) layer output, but the RELATIVE precision of tiny operands degrades.
 *   tests/test_gpu_parity.py::test_split_operand_range drives all three regimes against the oracle. */
typedef enum {
    QBOLD_ENC_F32 = 0,  /* float32-grade: operands split into two f16 halves, three MFMAs per tile */
    QBOLD_ENC_BF16 = 1  /* operands rounded to bfloat16, float32 accumulate (BASELINE config 5:
                           "bf16 forward / fp32 ELBO accum"); sampling and ELBO stay float32 */
} qbold_encoder_precision;

typedef enum {
    QBOLD_ACT_RELU = 0, /* 'relu': both of the reference's configuration files */
    QBOLD_ACT_GELU = 1  /* 'gelu' (Keras: exact, 0.5 x (1 + erf(x / sqrt 2))): the class default of EncoderTrainer */
} qbold_activation;

/* Voxel-wise encoder geometry (model.py:122-223; 3x3x1 convolutions act through their centre
 * tap on (N,1,1,1,T) voxel batches). */
typedef struct {
    int32_t T;                  /* no_ip_images */
    int32_t U;                  /* no_units */
    int32_t L;                  /* no_intermediate_layers */
    int32_t channelwise_gating; /* model.py:160-162 */
    float gate_offset;          /* model.py:169 */
    int32_t spatial_taps;       /* 1: Wr1/Wr2 hold the centre tap [U][U] only (voxel batches);
                                   9: full 3x3x1 kernels [3][3][U][U] in Keras order (model.py:152-157) */
    int32_t precision;          /* qbold_encoder_precision; the packed image (qbold_encoder_pack) and the
                                   kernels that read it must be given the same value */
    int32_t activation;         /* qbold_activation: EncoderTrainer's activation_type (model.py:60, 115-120, 151, 155).
                                   QBOLD_ACT_GELU runs on the layer-wise entry points (qbold_encoder_train_fwd / _bwd,
                                   qbold_encoder_spatial_fwd / _bwd: general kernels, pre-activations recomputed in the
                                   backward); the fused / wide entry points return QBOLD_ERR_UNSUPPORTED for it (ABI v4) */
    int32_t layer_norm;         /* EncoderTrainer's use_layer_norm (model.py:133-140): tfa GroupNormalization(groups = 1,
                                   axis = -1) in front of both activations of a block's residual path -- mean and biased
                                   variance over ALL positions and channels of one batch element (a voxel of a voxel batch,
                                   a whole crop of a crop batch), epsilon 1e-3, per-channel gamma / beta: 4 U more
                                   parameters per block behind the heads (qbold_encoder_num_params).  Layer-wise entry
                                   points only, like gelu (ABI v5) */
    float dropout_rate;         /* EncoderTrainer's dropout_rate: keras Dropout in front of each GroupNormalization (or
                                   activation).  Active in the TRAINING forward / backward when dropout_seed != 0: an
                                   element is dropped iff its 16-bit uniform of the library's Philox stream 5, keyed
                                   (dropout_seed; row, normalizer, column), is below rate 2^16; kept ones scale by
                                   1 / (1 - rate).  Inference (dropout_seed = 0) is the identity, as in Keras */
    uint64_t dropout_seed;      /* per-step seed, the same for a step's forward and backward; 0 = inference */
} qbold_encoder_shape;

/* Image-crop geometry of a [B][X][Y][Z][C] batch (train.py:17-72): voxel v = ((b X + x) Y + y) Z + z.
 * The 3x3x1 'same' convolutions couple (x, y) neighbours within one b and z. */
typedef struct {
    int32_t B, X, Y, Z;
} qbold_geometry;

typedef struct qbold_ctx qbold_ctx;

int qbold_abi_version(void);
const char* qbold_last_error(void);

/* ---- context ----------------------------------------------------------------------------- */
/* Replaces SignalGenerationLayer.__init__ (signals.py:18-53) + EncoderTrainer.__init__'s
 * se_idx (model.py:95).  Builds the tau grid, the folded float32 constants and the F(x) table
 * and uploads them to `device`. */
int qbold_ctx_create(const qbold_consts* consts, const qbold_loss_cfg* loss, int device,
                     qbold_ctx** out);
void qbold_ctx_destroy(qbold_ctx* ctx);
int qbold_ctx_num_taus(const qbold_ctx* ctx);
int qbold_ctx_se_idx(const qbold_ctx* ctx);
/* host_out: HOST float[T] */
int qbold_ctx_taus(const qbold_ctx* ctx, float* host_out);
int qbold_ctx_set_tissue_mode(qbold_ctx* ctx, int mode);
int qbold_ctx_tissue_mode(const qbold_ctx* ctx);
/* Gradient of the tissue integral: 1 (default) = TensorFlow's autodiff value, the J1-kernel
 * Simpson sum over all 129 nodes; 0 = the exact derivative of the float32 forward value, in which
 * node 0 is flat.  They differ by 1.95e-3 * x. */
int qbold_ctx_set_grad_node0(qbold_ctx* ctx, int on);
/* Which of several EQUIVALENT kernels an entry point dispatches to (default 0 = the fastest form of each).  Every bit
 * selects an older / simpler kernel that computes the same result; the tests use them to hold the fused kernels to
 * the layer-wise ones.  No bit skips work (the work-skipping hooks of the timing experiments exist only in
 * -DQBOLD_ABLATION builds, scripts/dev/build_ablation.sh). */
#define QBOLD_KSEL_RUNTIME_SE_IDX 4          /* ELBO kernels: run-time instead of compile-time spin-echo index */
#define QBOLD_KSEL_X_TABLE 8                 /* ELBO kernels: x-indexed F table per (draw, tau) instead of the per-tau OEF-indexed one */
#define QBOLD_KSEL_CONV_PER_TAP 256          /* 3x3x1 convolution as nine gathered GEMM launches */
#define QBOLD_KSEL_GENERAL_GEMM 512          /* layer GEMMs on the general xw_kernel */
#define QBOLD_KSEL_SEPARATE_GATE 2048        /* gate blend as its own launch; layer-wise backward: gate backward and the gating conv's backward-data product as two */
#define QBOLD_KSEL_SEPARATE_BWD_DATA 4096    /* a block's two backward-data GEMMs as two launches */
#define QBOLD_KSEL_SEPARATE_FORK 8192        /* skip / t of a block as two GEMMs */
#define QBOLD_KSEL_PER_HEAD_BWD 16384        /* one backward pass per head */
#define QBOLD_KSEL_PER_HEAD_FWD 32768        /* one forward GEMM per head */
#define QBOLD_KSEL_CONV_EXACT_F32 65536      /* 3x3x1 convolution on the f32-input MFMA */
#define QBOLD_KSEL_LAYERWISE_BWD 131072      /* voxel batches: layer-wise backward instead of the one-launch block kernels */
#define QBOLD_KSEL_BLOCK_BWD_SPLIT_DW 262144 /* block backward with separate weight-gradient launches */
#define QBOLD_KSEL_DW_EXACT_F32 524288      /* 3x3x1 weight gradients as exact f32 products instead of bf16 pieces */
#define QBOLD_KSEL_HEADS_BWD_LAYERWISE 1048576 /* heads' backward as delta tensor + GEMM + weight-gradient pass */
#define QBOLD_KSEL_SLAB_SUMS_SEPARATE 2097152 /* crop backward: every weight-gradient slab sum as its own launch instead of the queued ones */
#define QBOLD_KSEL_DW_BF16_PIECES 4194304     /* layer-wise weight gradients on three bfloat16 pieces per operand instead of two f16 halves */
#define QBOLD_KSEL_ALL (4 | 8 | 256 | 512 | 2048 | 4096 | 8192 | 16384 | 32768 | 65536 | 131072 | 262144 | 524288 | 1048576 | 2097152 | 4194304)
int qbold_ctx_set_kernel_selection(qbold_ctx* ctx, int mask);
/* Host-side evaluation of the uploaded table (for tests): F(x) and dF/dx, HOST arrays. */
int qbold_ctx_table_eval(const qbold_ctx* ctx, const float* host_x, float* host_F, float* host_dF,
                         int64_t n);

/* ---- forward signal model ------------------------------------------------------------------ */
/* SignalGenerationLayer.call without noise/misalignment (signals.py:55-114,137-138):
 * oef_dbv [V][2] -> signal [V][T]. */
int qbold_signal_fwd(const qbold_ctx* ctx, const float* oef_dbv, float* signal, int64_t V,
                     void* stream);
/* The same with the options configurations/optimal.yaml disables (signals.py:64-96).  hct [V]:
 * per-voxel haematocrit (variable_hct, :64-70) or NULL for the config's value.  Misalignment
 * (:80-96) with its random draws made by the caller: images t > from_index[v] are computed from
 * alt_oef_dbv[v] (the perturbed, clipped pair); from_index[v] >= T-1 leaves voxel v aligned.  Both
 * NULL: no misalignment. */
int qbold_signal_fwd_ex(const qbold_ctx* ctx, const float* oef_dbv, const float* hct,
                        const float* alt_oef_dbv, const int32_t* from_index, float* signal, int64_t V,
                        void* stream);
/* Vector-Jacobian product of the above: grad_signal [V][T] -> grad_oef_dbv [V][2]
 * (what tf.GradientTape yields through signals.py:55-114). */
int qbold_signal_bwd(const qbold_ctx* ctx, const float* oef_dbv, const float* grad_signal,
                     float* grad_oef_dbv, int64_t V, void* stream);

/* ---- encoder ------------------------------------------------------------------------------- */
/* Number of floats of the canonical (Keras-orientation, [in][out]) weight blob:
 * W0[T][U] b0[U] { Wc[U][U] bc[U] Wr1[taps][U][U] br1[U] Wr2[taps][U][U] br2[U] Wg[U][G] bg[G] } x L
 * Wf[U][5] bf[5] Ws[U][T] bs[T]   (model.py:181,144,152,156,164,196,211-214); taps = spatial_taps. */
int64_t qbold_encoder_num_params(const qbold_encoder_shape* shape);
/* Size in floats of the device workspace holding the MFMA-ordered copy of the weights. */
int64_t qbold_encoder_packed_floats(const qbold_encoder_shape* shape);
/* Re-order the canonical blob into the LDS image the kernels stage (call after every update). */
int qbold_encoder_pack(const qbold_ctx* ctx, const qbold_encoder_shape* shape,
                       const float* weights, float* packed, void* stream);
/* create_encoder forward, voxel-wise (model.py:97-113 normalise_data, :122-223):
 * x [N][T] -> out1 [N][5] (stream 1), out2 [N][5] (stream 2), sigma [N][T]; any may be NULL. */
int qbold_encoder_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed,
                      const float* x, float* out1, float* out2, float* sigma, int64_t N,
                      void* stream);

/* The same encoder for widths beyond the LDS-resident kernels (BASELINE config 3: 64 taus,
 * no_units 256): one weight-streaming split-f16 MFMA GEMM per layer over activations in HBM, with
 * bias / relu / gate / head fused into the epilogues.  Built for U = 128 or 256, L <= 8, T <= 64,
 * channel-wise gating, QBOLD_ENC_F32.  *_packed_floats / *_workspace_floats return the sizes (in
 * floats) of the image and of the caller-owned activation workspace for N voxels, or a negative
 * qbold_status.  stream_sel = 1: out_q = stream-1 parameters (model.py:199), out_log_sigma unused;
 * 2: out_q = stream-2 parameters (:208), out_log_sigma [N][T] = log of the sigma head (:211-214). */
int64_t qbold_encoder_wide_packed_floats(const qbold_encoder_shape* shape);
int64_t qbold_encoder_wide_workspace_floats(const qbold_encoder_shape* shape, int64_t N);
int qbold_encoder_wide_pack(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* weights,
                            float* packed, void* stream);
int qbold_encoder_wide_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed,
                           const float* x, int stream_sel, float* workspace, float* out_q,
                           float* out_log_sigma, int64_t N, void* stream);

/* The stream-2 encoder of the same wide shapes in ONE launch, activations resident in registers from the
 * first layer to the heads (HBM sees x once and the heads once; create_encoder, model.py:122-223).  Built for
 * U = 256, L = 1 or 2, T <= 16 or 49 <= T <= 64, channel-wise gating, QBOLD_ENC_F32; *_packed_floats returns
 * the image size in floats or a negative qbold_status for other shapes.  x [N][T] -> out_q [N][5] (model.py:208),
 * out_log_sigma [N][T] (the sigma head before its exp, :211-214).  packed, and x / out_log_sigma when T is a
 * multiple of 4, must be 16-byte aligned. */
int64_t qbold_encoder_fused_packed_floats(const qbold_encoder_shape* shape);
int qbold_encoder_fused_pack(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* weights,
                             float* packed, void* stream);
int qbold_encoder_fused_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed,
                            const float* x, float* out_q, float* out_log_sigma, int64_t N, void* stream);

/* ---- logit-Normal pieces -------------------------------------------------------------------- */
/* ReparamTrickLayer.call + forward_transform (model.py:15-50, 299-305): q [N][5], z [N][2] ->
 * oef_dbv [N][2]. */
int qbold_reparam(const qbold_ctx* ctx, const float* q, const float* z, float* oef_dbv, int64_t N,
                  void* stream);
/* logit_gaussian_mvg_log_prob (model.py:376-400 = logit_mvn.py:46-70): NEGATIVE log-density of
 * y [N][2] under params [N][5] -> out [N]. */
int qbold_logit_mvn_nlogp(const qbold_ctx* ctx, const float* y, const float* params, float* out,
                          int64_t N, void* stream);
/* kl_loss against a mixture-of-Gaussians population prior (use_mvg = False, use_population_prior = True,
 * mog_components = M > 1; model.py:666-685): q [N][5] (columns 0-3), comps DEVICE [M][4] raw parameters (M <= 16),
 * z = explicit normals [N][2] (OEF draw, DBV draw) or NULL for the in-kernel Philox stream -> kl [N] (unmasked). */
int qbold_kl_mog(const qbold_ctx* ctx, const float* q, const float* comps, int M, const float* z, uint64_t seed,
                 int64_t voxel0, float* kl, int64_t N, void* stream);
/* EncoderTrainer.squared_whitened_residual (model.py:423-441) = LogitMVN.squared_whitened_residual
 * (logit_mvn.py:20-38), a static method there and context-free here: obs, mean [N][2]; oef_log_std, dbv_log_std,
 * oef_dbv_cov [N] (already transformed) -> out [N] = || L^-1 (obs - mean) ||^2. */
int qbold_squared_whitened_residual(const float* obs, const float* mean, const float* oef_log_std,
                                    const float* dbv_log_std, const float* oef_dbv_cov, float* out, int64_t N,
                                    void* stream);
/* calculate_means(include_r2p=True, return_stds=True) (model.py:318-343): q [N][5] ->
 * means [N][3], vars [N][3] over n_samples reparameterised draws.  z = explicit normals
 * [N][n_samples][2] or NULL for the in-kernel Philox stream (seed, global voxel = voxel0+i). */
int qbold_posterior_moments(const qbold_ctx* ctx, const float* q, const float* z, int n_samples,
                            uint64_t seed, int64_t voxel0, float* means, float* vars, int64_t N,
                            void* stream);

/* EncoderTrainer.normalise_data (model.py:97-113): x [N][T] -> log(clip(x)/clip(x)[se]) [N][T]. */
int qbold_normalise(const qbold_ctx* ctx, const float* x, float* out, int64_t N, void* stream);

/* loglinear.fit_wls (loglinear.py:68-105): the log-linear comparator.  Per voxel a weighted
 * least-squares line ln S = c - R2' tau over the taus > tau_min (reference: 0.016) with weights
 * 1/tau, in closed form; DBV = c - ln S(tau = 0), OEF = R2' / (DBV gamma 4/3 pi dchi hct B0);
 * clipped to [0.01,0.8], [0.002,0.25], [1e-2,100].  signals [N][T] -> out [N][3] = (OEF, DBV, R2').
 * The taus are rounded to 7 decimals in float32 as the reference does (:126-127); fails with
 * QBOLD_ERR_ARG when no tau equals 0 or fewer than two taus exceed tau_min. */
int qbold_wls_fit(const qbold_ctx* ctx, const float* signals, double tau_min, float* out, int64_t N,
                  void* stream);

/* Element-wise parameter transforms of EncoderTrainer / LogitMVN (model.py:288-316 =
 * logit_mvn.py:72-100).  The FORWARD/BACKWARDS ops act on interleaved (OEF, DBV) pairs. */
typedef enum {
    QBOLD_TRANSFORM_STD = 0,            /* tanh(x)*3 - 1                     model.py:288-290 */
    QBOLD_TRANSFORM_OFFDIAG = 1,        /* tanh(x)*exp(-2)                   model.py:292-294 */
    QBOLD_INV_TRANSFORM_STD = 2,        /* atanh((x+1)/3)                    model.py:296-297 */
    QBOLD_FORWARD_TRANSFORM = 3,        /* sigmoid * range + min             model.py:299-305 */
    QBOLD_BACKWARDS_TRANSFORM = 4,      /* (y - min) / range                 model.py:307-311 */
    QBOLD_BACKWARDS_TRANSFORM_LOGIT = 5,/* ... followed by logit             model.py:312-314 */
    QBOLD_EXP = 6                       /* exp(x): sigma from the sigma head, model.py:214 */
} qbold_transform_op;
int qbold_transform(const qbold_ctx* ctx, int op, const float* in, float* out, int64_t n,
                    void* stream);

/* fine_tune_loss_fn(return_mean=False) before the mask multiply (model.py:527-563): data x [N][T],
 * mask [N] or NULL, predicted signals pred [N*S][T] and sigma [N*S][T] (row j belongs to voxel
 * j % N, the reference's S-fold batch tiling, model.py:529) -> nll [N*S]. */
int qbold_nll_fwd(const qbold_ctx* ctx, const float* x, const float* mask, const float* pred,
                  const float* sigma, float* nll, int64_t N, int S, void* stream);

/* mvg_kl_samples (model.py:592-610) per voxel: q, prior [N][5]; zk explicit normals [N][K][2] or
 * NULL for the Philox KL stream (seed, voxel0 + i) -> kl [N]. */
int qbold_kl_fwd(const qbold_ctx* ctx, const float* q, const float* prior, const float* zk, int K,
                 uint64_t seed, int64_t voxel0, float* kl, int64_t N, void* stream);
/* mvg_kl closed form, use_population_prior = False (model.py:612-652) -> kl [N]. */
int qbold_kl_closed(const qbold_ctx* ctx, const float* q, const float* prior, float* kl, int64_t N,
                    void* stream);
/* kl_loss of the DIAGONAL family (use_mvg = False, use_population_prior = False; model.py:686-721):
 * tfp LogitNormal.kl_divergence per dimension, closed form.  q, prior [N][5] (columns 0-3 used).
 * kl_v [N] or NULL; g_q [N][5] or NULL: [m_v > 0] d kl_v / d q is ADDED to columns 0-3 (analytic
 * KL: no stop-gradient); sums: DEVICE double[3] = (0, sum [m > 0] kl, sum m); workspace as
 * qbold_elbo_fwd. */
int qbold_kl_diag(const qbold_ctx* ctx, const float* q, const float* prior, const float* mask, float* kl_v,
                  float* g_q, double* sums, void* workspace, int64_t N, void* stream);

/* The counter-based normal stream the fused kernels consume: z [N][n][2] for global voxels
 * voxel0 .. voxel0+N-1; stream_id 0 = likelihood draws, 1 = KL draws, 2 = moments, 3 = noise.
 * Replaces tf.random.normal at model.py:25 with a reproducible, sharding-invariant generator
 * (Random123 Philox4x32-7, four draws per call: draw i = word i & 3 of call i >> 2 keyed (voxel, call, stream_id; seed);
 * Box-Muller on the word's low sixteen bits (radius, u1 = (lo + 0.5) 2^-16, so |z| <= 4.8549) and top 23 bits (angle,
 * (w >> 9) 2^-23 revolutions).  The definition is this library's; oracle/qbold_oracle.c restates it. */
int qbold_normals(const qbold_ctx* ctx, uint64_t seed, uint32_t stream_id, int64_t voxel0, int n,
                  float* z, int64_t N, void* stream);

/* Noise model of SignalGenerationLayer.call (signals.py:116-128), in place on signal [V][T]:
 * std[v][t] = mean_v(signal[.][t]) / (U(snr_lo, snr_hi)[v] * norm_snr[t]).  norm_snr: HOST [T]. */
int64_t qbold_noise_workspace_bytes(const qbold_ctx* ctx);
int qbold_signal_add_noise(const qbold_ctx* ctx, float* signal, const float* norm_snr_host,
                           float snr_lo, float snr_hi, uint64_t seed, int64_t voxel0,
                           void* workspace, int64_t V, void* stream);

/* ---- ELBO ----------------------------------------------------------------------------------- */
/* Size in bytes of the scratch the ELBO / fused kernels need for their per-workgroup partials. */
int64_t qbold_elbo_workspace_bytes(const qbold_ctx* ctx);

/* One voxel-ELBO evaluation given the encoder outputs: build_fine_tuner's sampling + forward
 * model (model.py:245-248,273), fine_tune_loss_fn (model.py:527-568), kl_loss -> mvg_kl_samples
 * (model.py:654-665, 592-610), masked sums as train.py:351.
 *   x [N][T], mask [N] (NULL = ones), q / prior [N][5], sigma [N][T]
 *   zs [N][S][2], zk [N][K][2] explicit normals, or NULL for the in-kernel Philox4x32-7 stream (qbold_normals)
 *   keyed (seed; global voxel voxel0+i; draw) -- identical for any sharding of the voxels
 *   nll_kl [N][2] per-voxel (nll averaged over S, kl) or NULL
 *   sums: DEVICE double[3] = (sum_v m*nll, sum_v [m>0] kl, sum_v m), overwritten
 *   workspace: qbold_elbo_workspace_bytes() bytes */
int qbold_elbo_fwd(const qbold_ctx* ctx, const float* x, const float* mask, const float* q,
                   const float* prior, const float* sigma, const float* zs, const float* zk, int S,
                   int K, uint64_t seed, int64_t voxel0, float* nll_kl, double* sums,
                   void* workspace, int64_t N, void* stream);

/* The same with the sigma head handed over BEFORE its exp (log_sigma [N][T], model.py:211-214), as the one-launch
 * wide encoder (qbold_encoder_fused_fwd) writes it; Philox stream only.  Built for the 64-tau protocol whose
 * tau = 0 image is index 12 (BASELINE config 3; table mode, Gaussian likelihood, one-image normalisation):
 * QBOLD_ERR_UNSUPPORTED otherwise. */
int qbold_elbo_fwd_logsigma(const qbold_ctx* ctx, const float* x, const float* mask, const float* q,
                            const float* prior, const float* log_sigma, int S, int K, uint64_t seed, int64_t voxel0,
                            float* nll_kl, double* sums, void* workspace, int64_t N, void* stream);

/* The whole hot path in one launch: encoder stream 2 (model.py:122-223) -> S reparameterised
 * samples -> forward model -> NLL, K-sample KL against `prior`, masked sums.  q_out [N][5]
 * receives the posterior parameters (NULL to skip); other arguments as qbold_elbo_fwd. */
/* Wide shapes (qbold_encoder_fused_packed_floats(shape) > 0 on the 64-tau protocol, BASELINE config 3) run as
 * two launches -- the one-launch encoder, then the ELBO kernel on its heads: `packed` is then the image of
 * qbold_encoder_fused_pack and `workspace` must hold qbold_vi_workspace_bytes(ctx, shape, N) bytes, 256-byte
 * aligned (for the LDS-resident shapes that is qbold_elbo_workspace_bytes()). */
int64_t qbold_vi_workspace_bytes(const qbold_ctx* ctx, const qbold_encoder_shape* shape, int64_t N);
int qbold_vi_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed,
                 const float* x, const float* mask, const float* prior, int S, int K, uint64_t seed,
                 int64_t voxel0, float* q_out, float* nll_kl, double* sums, void* workspace,
                 int64_t N, void* stream);

/* ---- gradients (training) ------------------------------------------------------------------- */
/* Adjoint of qbold_elbo_fwd with respect to the encoder's head outputs: what TensorFlow autodiff
 * yields through build_fine_tuner's sampling + fine_tune_loss_fn + kl_loss (model.py:239-286,
 * 527-568, 592-610; q is stop-gradient inside log q, model.py:596).
 *   log_sigma [N][T] (the sigma head BEFORE exp, model.py:211-214)
 *   g_q [N][5], g_log_sigma [N][T]:  m_v * d nll_v/d. + [m_v > 0] * d kl_v/d.   (NOT divided by
 *   sum(m): the caller scales the weight gradient once)
 *   nll_kl, sums, workspace: as qbold_elbo_fwd (same Philox stream -> same loss values).
 * Built for the full signal model in table mode; Gaussian or Student-t likelihood, linear or log
 * data, one- or three-image normalisation (the switches of qbold_loss_cfg). */
int qbold_elbo_bwd(const qbold_ctx* ctx, const float* x, const float* mask, const float* q,
                   const float* prior, const float* log_sigma, int S, int K, uint64_t seed,
                   int64_t voxel0, float* g_q, float* g_log_sigma, float* nll_kl, double* sums,
                   void* workspace, int64_t N, void* stream);

/* Encoder training (TensorFlow autodiff through create_encoder, model.py:122-223, in the Keras fit
 * loops train.py:285-376 / :379-427).  `weights` is the canonical blob; `workspace` holds
 * qbold_train_workspace_floats(shape, N) floats (saved activations + scratch).
 *   stream_sel 1: pre-training stream (out_q = output 0, model.py:199), 2: fine-tuning stream
 *   (out_q = second_net, out_log_sigma = sigma head before exp, model.py:208-214). */
int64_t qbold_train_workspace_floats(const qbold_encoder_shape* shape, int64_t N);
int qbold_encoder_train_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* weights,
                            const float* x, int stream_sel, float* workspace, float* out_q,
                            float* out_log_sigma, int64_t N, void* stream);
/* stream_sel = 2 of the call above in ONE launch for the shapes of the LDS-resident kernels (U <= 64, L <= 2,
 * T = 11 or 24, channel-wise gating, QBOLD_ENC_F32; 0 < N < 2^23): `packed` is the image of
 * qbold_encoder_pack; the workspace receives the same saved tensors, so qbold_encoder_train_bwd follows
 * unchanged.  save_all = 2 saves every tensor; 1 leaves out the two per block (skip, gate logits) and 0 the four
 * per block (also t, r) that qbold_encoder_train_bwd recomputes -- pass 2 - qbold_encoder_train_bwd_recomputes(ctx,
 * shape, N), or 2.  Products are the float32-grade split-f16 ones of qbold_encoder_fwd (operand range above: heads
 * come out NaN beyond it), not the exact-f32 GEMMs of qbold_encoder_train_fwd. */
int qbold_encoder_train_fwd_fused(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* packed,
                                  const float* x, int save_all, float* workspace, float* out_q,
                                  float* out_log_sigma, int64_t N, void* stream);
/* What qbold_encoder_train_bwd (stream 2, voxel batch) recomputes from each gated block's input instead of
 * reading it: 0 nothing (the layer-wise backward), 1 skip and the gate logits (one launch per block for the data
 * side), 2 also t and r (the same launch accumulates the block's weight gradients). */
int qbold_encoder_train_bwd_recomputes(const qbold_ctx* ctx, const qbold_encoder_shape* shape, int64_t N);
/* The same on image crops: geom->B*X*Y*Z voxels, stream 2 with its 3x3x1 'same' convolutions
 * (shape->spatial_taps must be 9).  geom = NULL is the voxel-batch call above. */
int qbold_encoder_spatial_fwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* weights,
                              const float* x, const qbold_geometry* geom, float* workspace, float* out_q,
                              float* out_log_sigma, void* stream);
int qbold_encoder_spatial_bwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* weights,
                              const qbold_geometry* geom, float* workspace, const float* g_q,
                              const float* g_log_sigma, const double* sums, float* grad, void* stream);
/* smoothness_loss (model.py:726-754) on crops: q [V][5], mask [V].  tv_sum: DEVICE double[1] =
 * sum |dx| + sum |dy| (divide by sum(mask)); if g_q is not NULL, weight * d tv_sum / d q is ADDED to
 * g_q [V][5] (unnormalised, like qbold_elbo_bwd's output). */
int qbold_smoothness(const qbold_ctx* ctx, const float* q, const float* mask, const qbold_geometry* geom,
                     float weight, float* g_q, double* tv_sum, void* stream);
/* g_q [N][5], g_log_sigma [N][T] (stream 2; may be NULL): gradients of the loss with respect to
 * the head outputs.  sums: DEVICE double[3] whose [2] is sum(mask) -- the head gradients are
 * divided by it (NULL: already normalised).  grad: canonical layout, overwritten.
 * When qbold_encoder_train_bwd_recomputes(...) is 2 the backward RECOMPUTES each block (skip, t, r, gate logits) with
 * the split-f16 products of qbold_encoder_train_fwd_fused and takes its relu masks from that recomputation: it is the
 * adjoint of THAT forward.  After the exact-f32 layer-wise qbold_encoder_train_fwd the masks can differ for
 * activations within ~1e-6 of zero (where the relu's derivative is a convention anyway); callers that need the adjoint
 * of the layer-wise forward bit for bit select QBOLD_KSEL_LAYERWISE_BWD.
 * Operand range of the layer-wise backward (crops, and voxel batches under QBOLD_KSEL_LAYERWISE_BWD): the 3x3x1
 * backward-data products and the weight gradients run on split-f16 operands.  Deltas are NOT bound by f16's range:
 * with `sums` given the backward-data products lift them by 2^floor(log2 sums[2]) (exact), and the weight-gradient
 * kernels keep a running power-of-two scale per wave that follows the largest |delta| seen (accumulators rescaled,
 * exact), whatever the caller's normalisation.  Activations are taken as they are: the forward's limit above applies
 * (an activation beyond 65504 gives non-finite gradients, never a clamp), and an activation below 2^-14 in magnitude
 * keeps an absolute 2^-25 in the weight gradients.  QBOLD_KSEL_DW_BF16_PIECES (three bfloat16 pieces per operand, no
 * range at all) and QBOLD_KSEL_DW_EXACT_F32 select the other forms. */
int qbold_encoder_train_bwd(const qbold_ctx* ctx, const qbold_encoder_shape* shape, const float* weights,
                            int stream_sel, float* workspace, const float* g_q, const float* g_log_sigma,
                            const double* sums, float* grad, int64_t N, void* stream);
/* synthetic_data_loss (model.py:449-514; use_mvg, no r2p term) per voxel and its gradient: y_true
 * rows of ld_y floats (OEF, DBV first), q [N][5] -> loss_v [N] (may be NULL), g_q [N][5] =
 * scale * d loss_v / d q.  inv_gamma_alpha * inv_gamma_beta > 0 adds the inverse-gamma prior on the
 * two marginal variances (:492-507: - log IG(exp(s_o)^2) - log IG(exp(s_d)^2 + q[4]^2)). */
int qbold_synth_loss_bwd(const qbold_ctx* ctx, const float* y_true, int ld_y, const float* q, float* g_q,
                         float* loss_v, float scale, double inv_gamma_alpha, double inv_gamma_beta,
                         int64_t N, void* stream);
/* The learned inverse-gamma hyper-prior of synthetic_data_loss (infer_inv_gamma with the diagonal family,
 * model.py:201-205, 454-455, 493-507): ig_host = HOST double[4] (alpha_oef, beta_oef, alpha_dbv, beta_dbv), the
 * exp-activated VariableLayer values.  Per voxel - log IG(exp(2 s_o); a_o, b_o) - log IG(exp(2 s_d); a_d, b_d) is
 * ADDED to loss_v [N], scale * its gradient to g_q [N][5] (columns 1, 3); stats: DEVICE double[4] = sum log v_o,
 * sum 1 / v_o, sum log v_d, sum 1 / v_d -- what the gradient with respect to the four hyper-parameters needs
 * (d / d a = digamma(a) - log b + mean log v, d / d b = - a / b + mean 1 / v).  Any of g_q / loss_v / stats may be
 * NULL.  Call after qbold_synth_loss_bwd. */
int qbold_hyper_prior_bwd(const qbold_ctx* ctx, const float* q, const double* ig_host, float scale, float* g_q,
                          float* loss_v, double* stats, int64_t N, void* stream);
/* The R2' term of synthetic_data_loss (use_r2p_loss, model.py:475-490): n_samples (reference: 10)
 * reparameterised draws of (OEF, DBV) from q [N][5], r = dw(OEF) DBV (calculate_r2p :524-525), a normal
 * fitted by the draws' mean and biased std, gaussian_nll (:403-404) of the true R2' y_true[:, 2] under it.
 * ADDS the per-voxel value to loss_v [N] and scale * d / d q to g_q [N][5] (either may be NULL): call it
 * after qbold_synth_loss_bwd.  z = explicit normals [N][n_samples][2] or NULL for the in-kernel Philox
 * stream (seed, global voxel = voxel0 + i). */
int qbold_r2p_loss_bwd(const qbold_ctx* ctx, const float* y_true, int ld_y, const float* q, const float* z,
                       int n_samples, uint64_t seed, int64_t voxel0, float scale, float* g_q, float* loss_v,
                       int64_t N, void* stream);
/* One tfa.optimizers.AdamW step (train.py:308-310, 382-385) on a flat blob: decoupled decay
 * var -= weight_decay * var, Keras Adam moments and bias correction at step t >= 1, eps 1e-7. */
int qbold_adamw_step(const qbold_ctx* ctx, float* params, const float* grads, float* m, float* v,
                     int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                     int64_t t, void* stream);

#ifdef __cplusplus
}
#endif
#endif

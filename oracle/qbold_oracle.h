/*
 * qbold_oracle.h -- CPU restatement ("oracle") of the qBOLD-VI voxel-wise ELBO hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path (qbold_vi_amd/, include/,
 * train.py, qbold_train_model.py) may include, link, import or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PARITY UNPINNED: the reference (wearepal/qBOLD-VI) ships no tests, golden vectors or
 * known-answer fixtures for this path, and its arithmetic lives in TensorFlow >=2.5 / TFP >=0.13
 * (requirements.txt:3,9; unpinned floors, not vendored, not installable here).  This file is a
 * line-by-line restatement of the reference's Python (file:line cited at each function) with
 * TensorFlow's float32 kernels restated from their published algorithms (Cephes j0f for
 * tf.math.special.bessel_j0, Random123 Philox4x32-10 for the counter RNG).  It is pinned only by
 * independent cross-checks (scipy f64 quadrature / Bessel, closed-form KL, Random123 KATs); see
 * tests/test_oracle_*.py and DESIGN.md.
 *
 * The same source builds in float32 (REAL=float, the parity target) and float64 (REAL=double,
 * "truth" for error budgets) -- see oracle/Makefile.
 */
#ifndef QBOLD_ORACLE_H
#define QBOLD_ORACLE_H

#include <stdint.h>

#ifndef QBO_REAL
#define QBO_REAL float
#endif
typedef QBO_REAL real;

#define QBO_MAX_T 64

#ifdef __cplusplus
extern "C" {
#endif

/* Physics/system constants: reference `config` INI [DEFAULT] (config:1-38) as parsed by
 * SignalGenerationLayer.__init__ (signals.py:18-53). */
typedef struct {
    double gamma, b0, dchi, te, r2t, tr, ti, t1b, hct;
    double tau_start, tau_end, tau_step;
    int32_t full_model;    /* signals.py:192 */
    int32_t include_blood; /* signals.py:100 */
} qbo_phys;

/* NLL / normalisation switches of EncoderTrainer (model.py:54-95). */
typedef struct {
    int32_t se_idx;                    /* model.py:95 */
    int32_t multi_image_normalisation; /* model.py:102,540 */
    int32_t predict_log_data;          /* model.py:547 */
    int32_t use_student_t;             /* model.py:557: df is not None and df < 50 */
    double student_t_df;
} qbo_loss_cfg;

/* Voxel-wise encoder weights (model.py:122-223 with 3x3x1 convs reduced to their centre tap,
 * exact for (N,1,1,1,T) inputs with 'same' padding).  Keras kernel orientation [in][out]. */
typedef struct {
    int32_t T, U, L;
    int32_t channelwise_gating; /* model.py:160-162 */
    int32_t taps;               /* 1: Wr1/Wr2 are centre taps [L][U][U]; 9: full 3x3x1 kernels
                                   [L][3][3][U][U] (Keras kernel layout, model.py:152-157) */
    double gate_offset;         /* model.py:169 */
    const real *W0, *b0;        /* [T][U], [U]            model.py:181 */
    const real *Wc, *bc;        /* [L][U][U], [L][U]      model.py:144 (shared by both streams) */
    const real *Wr1, *br1;      /* [L][U][U], [L][U]      model.py:152 centre tap */
    const real *Wr2, *br2;      /* [L][U][U], [L][U]      model.py:156 centre tap */
    const real *Wg, *bg;        /* [L][U][G], [L][G]      model.py:164, G = U or 1 */
    const real *Wf, *bf;        /* [U][5], [5]            model.py:196 (shared by both streams) */
    const real *Ws, *bs;        /* [U][T], [T]            model.py:211-214 */
} qbo_weights;

int qbo_real_bytes(void);
void qbo_set_threads(int n);
/* float64 test hook: evaluate F(x) with Simpson node 0 removed (float32 reference semantics). */
void qbo_set_node0_zero(int on);

/* test hook: round both operands of every encoder dense product to bfloat16 (voxel-wise encoder). */
void qbo_set_encoder_bf16(int on);
/* test hook: add_normalizer (model.py:131-140) in qbo_encoder_fwd_spatial: GroupNormalization parameters ln [L][4][U]
 * (gamma1, beta1, gamma2, beta2 per block; NULL = off) and a training-mode dropout (rate, seed; seed 0 = inference) */
void qbo_set_normalizer(const real *ln, double dropout_rate, uint64_t dropout_seed);
/* test hook: 'gelu' (Keras exact form) instead of 'relu' in the encoder restatements (model.py:60, 115-120) */
void qbo_set_activation_gelu(int on);

/* tf.math.special.bessel_j0 for float32 = Eigen generic_j0<float> = Cephes j0f. */
real qbo_j0(real x);
real qbo_j1(real x);
void qbo_j0_array(const real *x, real *y, int64_t n);

/* tf.range(tau_start, tau_end, tau_step, float32) -- signals.py:34-35.  Returns T. */
int qbo_taus(const qbo_phys *P, real *taus);

/* Simpson-129 tissue integral F(x), x = tau*dw -- signals.py:159-185. */
real qbo_tissue_F(real x);
void qbo_tissue_F_array(const real *x, real *F, int64_t n);
/* dF/dx with the J1 kernel (what TF autodiff yields through bessel_j0). */
real qbo_tissue_dF(real x);

/* SignalGenerationLayer.call without noise/misalignment -- signals.py:55-114,137-138. */
void qbo_signal_fwd(const qbo_phys *P, const real *oef_dbv /*[V][2]*/, real *signal /*[V][T]*/,
                    int64_t V);
/* The same with the options optimal.yaml disables (signals.py:64-96): hct [V] per-voxel haematocrit
 * (variable_hct) or NULL; misalignment with explicit draws -- images t > from_idx[v] are computed
 * from alt[v] ([V][2], the perturbed and clipped (OEF, DBV)); from_idx >= T-1 or NULL: aligned. */
void qbo_signal_fwd_ex(const qbo_phys *P, const real *oef_dbv, const real *hct, const real *alt,
                       const int32_t *from_idx, real *signal, int64_t V);
/* d signal[v][t] / d (oef, dbv): jac [V][T][2]. */
void qbo_signal_jac(const qbo_phys *P, const real *oef_dbv, real *jac, int64_t V);

/* EncoderTrainer.normalise_data -- model.py:97-113. */
void qbo_normalise(const qbo_loss_cfg *C, const real *x /*[N][T]*/, real *n /*[N][T]*/, int T,
                   int64_t N);
/* create_encoder forward, voxel-wise -- model.py:122-223.  Any output may be NULL. */
void qbo_encoder_fwd(const qbo_weights *W, const qbo_loss_cfg *C, const real *x /*[N][T]*/,
                     real *out1 /*[N][5]*/, real *out2 /*[N][5]*/, real *sigma /*[N][T]*/,
                     int64_t N);

/* create_encoder forward on image volumes x [B][X][Y][Z][T] with the 3x3x1 'same'-padded
 * convolutions of stream 2 (model.py:152-157); W->taps must be 9.  out2 [B X Y Z][5],
 * sigma [B X Y Z][T]. */
void qbo_encoder_fwd_spatial(const qbo_weights *W, const qbo_loss_cfg *C, const real *x, int B, int X,
                             int Y, int Z, real *out2, real *sigma);
/* smoothness_loss (model.py:726-754) numerator: sum |dx| + sum |dy| of the forward-transformed,
 * range-scaled means over neighbour pairs whose masks are both > 0 (divide by sum(mask)). */
double qbo_smoothness_sum(const real *q /*[V][5]*/, const real *mask /*[V]*/, int B, int X, int Y,
                          int Z);

/* ReparamTrickLayer.call (use_mvg) + forward_transform -- model.py:24-31,47-50,299-305. */
void qbo_reparam(const real *q /*[N][5]*/, const real *z /*[N][2]*/, real *oef_dbv /*[N][2]*/,
                 int64_t N);
/* logit_gaussian_mvg_log_prob: NEGATIVE log-density -- model.py:376-400 = logit_mvn.py:46-70. */
void qbo_logit_mvn_nlogp(const real *y /*[N][2]*/, const real *p /*[N][5]*/, real *out /*[N]*/,
                         int64_t N);
/* synthetic_data_loss (use_mvg, no r2p loss, no inv-gamma) -- model.py:449-514. */
double qbo_synthetic_data_loss(const real *y_true /*[N][3]*/, const real *q /*[N][5]*/, int64_t N);
/* ... plus the fixed inverse-gamma prior on the marginal variances -- model.py:492-507. */
double qbo_synthetic_data_loss_ig(const real *y_true, const real *q, double alpha, double beta, int64_t N);

/* fine_tune_loss_fn per voxel (return_mean=False, before masking) -- model.py:527-563. */
void qbo_nll(const qbo_loss_cfg *C, const real *x /*[N][T]*/, const real *mask /*[N]*/,
             const real *pred /*[N][T]*/, const real *sigma /*[N][T]*/, real *nll /*[N]*/, int T,
             int64_t N);
/* mvg_kl_samples per voxel with explicit normals z [N][K][2] -- model.py:592-610. */
void qbo_kl_samples(const real *q, const real *prior, const real *z, int K, real *kl /*[N]*/,
                    int64_t N);
/* mvg_kl closed form (use_population_prior=False) -- model.py:612-652. */
void qbo_kl_closed(const real *q, const real *prior, real *kl /*[N]*/, int64_t N);
/* Diagonal family (use_mvg = False): closed-form kl_loss per voxel (model.py:686-716) and
 * logit_gaussian_log_prob (:406-421); rows 5 wide, columns 0-3 used. */
void qbo_kl_diag(const real *q, const real *prior, real *kl, int64_t N);
double qbo_population_prior_cost(const real *prior4 /*[4]*/, int batch);
void qbo_kl_mog(const real *q /*[N][5]*/, const real *comps /*[M][4]*/, int M, const real *z /*[N][2]*/, real *kl, int64_t N);
void qbo_logit_gaussian_nlogp(const real *y /*[N][2]*/, const real *p /*[N][5]*/, real *out, int64_t N);
/* calculate_means(include_r2p=True, return_stds=True) with explicit z [N][n][2] --
 * model.py:318-343.  means/vars [N][3] = (OEF, DBV, R2'). */
void qbo_moments(const qbo_phys *P, const real *q, const real *z, int n, real *means, real *vars,
                 int64_t N);

/* Whole voxel-ELBO evaluation: per voxel nll (mean over S) and kl, masked sums.
 * zs [N][S][2], zk [N][K][2].  sums[3] = (sum m*nll, sum where(m>0,kl), sum m) in double. */
void qbo_elbo(const qbo_phys *P, const qbo_loss_cfg *C, const real *x, const real *mask,
              const real *q, const real *prior, const real *sigma, const real *zs, int S,
              const real *zk, int K, real *nll_v, real *kl_v, double *sums, int64_t N);

/* Random123 Philox4x32-10. */
void qbo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void qbo_philox4x32_7(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* Counter-RNG normals shared bit-for-bit (integer part) with the HIP kernels:
 * ctr = (voxel_lo, voxel_hi, draw>>1 pair index, stream), key = (seed_lo, seed_hi).
 * z [N][n][2] for global voxels voxel0 .. voxel0+N-1. */
void qbo_philox_normals(uint64_t seed, uint32_t stream, int64_t voxel0, int64_t N, int n, real *z);

#ifdef __cplusplus
}
#endif
#endif

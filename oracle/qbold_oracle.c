/*
 * qbold_oracle.c -- CPU restatement of the qBOLD-VI voxel-wise ELBO hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see qbold_oracle.h for the full statement).
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * repository root).  Arithmetic is carried out in `real` (float for the parity build) in the
 * reference's order of operations; Python-side double constants are folded in double and cast to
 * `real` exactly where TensorFlow would convert them to a float32 tensor.
 */
#define _GNU_SOURCE
#include "qbold_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define R(x) ((real)(x))

#if defined(QBO_DOUBLE)
#define r_exp exp
#define r_log log
#define r_sqrt sqrt
#define r_tanh tanh
#define r_cos cos
#define r_fabs fabs
#define r_lgamma lgamma
#define r_log1p log1p
#else
#define r_exp expf
#define r_log logf
#define r_sqrt sqrtf
#define r_tanh tanhf
#define r_cos cosf
#define r_fabs fabsf
#define r_lgamma lgammaf
#define r_log1p log1pf
#endif

int qbo_real_bytes(void) { return (int)sizeof(real); }

void qbo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Bessel J0 / J1.  tf.math.special.bessel_j0 (signals.py:170) dispatches to Eigen's
 * generic_j0<T,float>, which is Cephes single-precision j0f.c (Moshier): rational/polynomial
 * in z=x^2 on [0,2] (with the 1 - z/4 shortcut below 1e-3), modulus/phase asymptotic form above.
 * The float64 build uses libm's j0/j1 as "truth".
 * ---------------------------------------------------------------------------------------- */
#if !defined(QBO_DOUBLE)
static inline float polevl(float x, const float *c, int n) {
    float r = c[0];
    for (int i = 1; i <= n; ++i) r = r * x + c[i];
    return r;
}

static const float J0_MO[8] = {-6.838999669318810E-002f, 1.864949361379502E-001f,
                               -2.145007480346739E-001f, 1.197549369473540E-001f,
                               -3.560281861530129E-003f, -4.969382655296620E-002f,
                               -3.355424622293709E-006f, 7.978845717621440E-001f};
static const float J0_PH[8] = {3.242077816988247E+001f,  -3.630592630518434E+001f,
                               1.756221482109099E+001f,  -4.974978466280903E+000f,
                               1.001973420681837E+000f,  -1.939906941791308E-001f,
                               6.490598792654666E-002f,  -1.249992184872738E-001f};
static const float J0_JP[5] = {-6.068350350393235E-008f, 6.388945720783375E-006f,
                               -3.969646342510940E-004f, 1.332913422519003E-002f,
                               -1.729150680240724E-001f};
#define J0_DR1 5.78318596294678452118f
#define PIO4F 0.7853981633974483096f

real qbo_j0(real xx) {
    float x = fabsf(xx);
    if (x <= 2.0f) {
        float z = x * x;
        if (x < 1.0e-3f) return 1.0f - 0.25f * z;
        return (z - J0_DR1) * polevl(z, J0_JP, 4);
    }
    float q = 1.0f / x;
    float w = 1.0f / sqrtf(x); /* Eigen: prsqrt(y); Cephes: sqrtf(q) */
    float p = w * polevl(q, J0_MO, 7);
    w = q * q;
    float xn = q * polevl(w, J0_PH, 7) - PIO4F;
    return p * cosf(xn + x);
}

static const float J1_JP[5] = {-4.878788132172128E-009f, 6.009061827883699E-007f,
                               -4.541343896997497E-005f, 1.937383947804541E-003f,
                               -3.405537384615824E-002f};
static const float J1_MO[8] = {6.913942741265801E-002f,  -2.284801500053359E-001f,
                               3.138238455499697E-001f,  -2.102302420403875E-001f,
                               5.435364690523026E-003f,  1.493389585089498E-001f,
                               4.976029650847191E-006f,  7.978845453073848E-001f};
static const float J1_PH[8] = {-4.497014141919556E+001f, 5.073465654089319E+001f,
                               -2.485774108720340E+001f, 7.222973196770240E+000f,
                               -1.544842782180211E+000f, 3.503787691653334E-001f,
                               -1.637986776941202E-001f, 3.749989509080821E-001f};
#define J1_Z1 1.46819706421238932572E1f
#define THPIO4F 2.35619449019234492885f

real qbo_j1(real xx) {
    float x = fabsf(xx);
    float r;
    if (x <= 2.0f) {
        float z = x * x;
        r = (z - J1_Z1) * x * polevl(z, J1_JP, 4);
    } else {
        float q = 1.0f / x;
        float w = sqrtf(q);
        float p = w * polevl(q, J1_MO, 7);
        w = q * q;
        float xn = q * polevl(w, J1_PH, 7) - THPIO4F;
        r = p * cosf(xn + x);
    }
    return xx < 0 ? -r : r;
}
#else
real qbo_j0(real x) { return j0(x); }
real qbo_j1(real x) { return j1(x); }
#endif

void qbo_j0_array(const real *x, real *y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = qbo_j0(x[i]);
}

/* ------------------------------------------------------------------------------------------
 * tau grid -- signals.py:34-35: tf.range(start, end, step, dtype=float32).
 * size = ceil(|end-start|/|step|) (TF RangeSize, computed in double from the float32-cast
 * limits); element i = start + i*step in float32 (multiply form; the accumulate form used by
 * older TF differs by <= 1 ulp at a few i).
 * ---------------------------------------------------------------------------------------- */
int qbo_taus(const qbo_phys *P, real *taus) {
    real s = R(P->tau_start), e = R(P->tau_end), d = R(P->tau_step);
    int T = (int)ceil(fabs((double)e - (double)s) / fabs((double)d));
    if (T > QBO_MAX_T) T = QBO_MAX_T;
    for (int i = 0; i < T; ++i) taus[i] = s + R(i) * d;
    return T;
}

/* ------------------------------------------------------------------------------------------
 * Tissue integral -- signals.py:159-185 (compose / integral).
 *   int_parts = tf.linspace(1e-5, 1, 129)                                       :166-168
 *   y = (2+u)*sqrt(1-u)*(1 - J0(1.5*(tau*dw)*u)) / (3*u^2)                       :169-171
 *   Simpson: (y_a + y_b + 4 y_m) * (h/3), h = (u[2]-u[0])/2, reduce_sum          :180-185
 * tf.linspace float32: start + i*delta with delta=(stop-start)/(num-1), last element = stop.
 * ---------------------------------------------------------------------------------------- */
#define NNODE 129
static real g_u[NNODE];
static real g_pre[NNODE]; /* (2+u)*sqrt(1-u) */
static real g_den[NNODE]; /* 3*u^2 */
static real g_h3;
static int g_nodes_ready = 0;

static void init_nodes(void) {
    if (g_nodes_ready) return;
#pragma omp critical(qbo_nodes)
    {
        if (!g_nodes_ready) {
            real a = R(1e-5), b = R(1);
            real delta = (b - a) / R(NNODE - 1);
            for (int i = 0; i < NNODE; ++i) g_u[i] = a + R(i) * delta;
            g_u[NNODE - 1] = b;
            for (int i = 0; i < NNODE; ++i) {
                g_pre[i] = (R(2) + g_u[i]) * r_sqrt(R(1) - g_u[i]);
                g_den[i] = R(3.0) * (g_u[i] * g_u[i]);
            }
            real h = (g_u[2] - g_u[0]) / R(2.0);
            g_h3 = h / R(3.0);
            g_nodes_ready = 1;
        }
    }
}

/* Test hook for the float64 build: drop node 0 from F, i.e. the value the float32 reference
 * computes (1 - J0 rounds to 0 there) evaluated without float32 noise. */
static int g_node0_zero = 0;
void qbo_set_node0_zero(int on) { g_node0_zero = on; }

static inline real node_y(real x, int i) {
    if (i == 0 && g_node0_zero) return 0;
    real arg = (R(1.5) * x) * g_u[i];
    return g_pre[i] * (R(1.0) - qbo_j0(arg)) / g_den[i];
}

real qbo_tissue_F(real x) {
    init_nodes();
    real y[NNODE];
    for (int i = 0; i < NNODE; ++i) y[i] = node_y(x, i);
    real acc = 0;
    for (int m = 0; m < (NNODE - 1) / 2; ++m)
        acc += (y[2 * m] + y[2 * m + 2] + R(4.0) * y[2 * m + 1]) * g_h3;
    return acc;
}

void qbo_tissue_F_array(const real *x, real *F, int64_t n) {
    init_nodes();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) F[i] = qbo_tissue_F(x[i]);
}

/* d/dx of the Simpson sum: d/dx[1-J0(1.5 x u)] = 1.5 u J1(1.5 x u). */
real qbo_tissue_dF(real x) {
    init_nodes();
    real y[NNODE];
    for (int i = 0; i < NNODE; ++i) {
        real arg = (R(1.5) * x) * g_u[i];
        y[i] = g_pre[i] * (R(1.5) * g_u[i] * qbo_j1(arg)) / g_den[i];
    }
    real acc = 0;
    for (int m = 0; m < (NNODE - 1) / 2; ++m)
        acc += (y[2 * m] + y[2 * m + 2] + R(4.0) * y[2 * m + 1]) * g_h3;
    return acc;
}

/* ------------------------------------------------------------------------------------------
 * Forward signal model -- signals.py:55-114 (no misalignment, no noise), calc_tissue :152-209,
 * calc_blood :233-247, calculate_dw_static :142-144.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int T;
    real taus[QBO_MAX_T];
    real dw_coef;  /* (4/3) pi gamma b0 dchi hct                signals.py:144 */
    real dw_coef_nohct; /* the same without hct (variable_hct, signals.py:64-70) */
    real e_te_r2t; /* exp(-te*r2t)                              signals.py:172 */
    real m_bld_nb; /* m_bld * nb                                signals.py:102-107 */
    real g0_c1;    /* (4/45) hct (1-hct)                        signals.py:239 */
    real g0_c2;    /* 4 pi b0 dchi                              signals.py:239 */
    real half_g2;  /* 0.5*gamma^2                               signals.py:241 */
    real td2;      /* td^2                                      signals.py:241 */
    real e_r2b_te; /* exp(-r2b*te)                              signals.py:241 */
    real blood_B[QBO_MAX_T]; /* bracket of signals.py:242-247, per tau */
    real r2t_te;
} fwd_consts;

static void make_consts(const qbo_phys *P, fwd_consts *c) {
    c->T = qbo_taus(P, c->taus);
    c->dw_coef = R((4.0 / 3.0) * M_PI * P->gamma * P->b0 * P->dchi * P->hct);
    c->dw_coef_nohct = R((4.0 / 3.0) * M_PI * P->gamma * P->b0 * P->dchi);
    c->e_te_r2t = r_exp(R(-P->te * P->r2t));
    c->r2t_te = R(-P->r2t * P->te);
    /* m_bld: tf.math.exp of python floats -> float32 scalars          signals.py:105 */
    real e1 = r_exp(R(-(P->tr - P->ti) / P->t1b));
    real e2 = r_exp(R(-P->ti / P->t1b));
    real m_bld = R(1) - (R(2) - e1) * e2;
    c->m_bld_nb = m_bld * R(0.775);
    c->g0_c1 = R((4.0 / 45.0) * P->hct * (1.0 - P->hct));
    c->g0_c2 = R(4.0 * M_PI * P->b0 * P->dchi);
    c->half_g2 = R(0.5 * (P->gamma * P->gamma));
    double r2b = 1.0 / 0.189;
    double td = (pow(2.6, 2.0) / 2.0) * 1e-3;
    c->td2 = R(td * td);
    c->e_r2b_te = r_exp(R(-r2b * P->te));
    real te = R(P->te), tdr = R(td);
    real te_td = R(P->te / td);
    real s0 = r_sqrt(R(0.25 + (P->te / td)));
    for (int t = 0; t < c->T; ++t) {
        real a = r_sqrt(R(0.25) + ((te + c->taus[t]) / tdr));
        real b = r_sqrt(R(0.25) + ((te - c->taus[t]) / tdr));
        c->blood_B[t] = te_td + s0 + R(1.5) - (R(2.0) * a) - (R(2.0) * b);
    }
}

/* hct < 0: the fixed haematocrit of the config (python float folded into the constants,
 * signals.py:46,78); otherwise a per-voxel float32 tensor (variable_hct, signals.py:64-70), which
 * changes where the float64 constants are rounded: ((K * hct) * oef), ((4/45 * hct) * (1 - hct)). */
static inline void signal_one_h(const qbo_phys *P, const fwd_consts *c, real oef, real dbv, real hct,
                                real *out) {
    real dw = hct < 0 ? c->dw_coef * oef : (c->dw_coef_nohct * hct) * oef;
    real g0_c1 = hct < 0 ? c->g0_c1 : (R(4.0 / 45.0) * hct) * (R(1) - hct);
    real bw;
    real g = 0;
    if (P->include_blood) {
        bw = c->m_bld_nb * dbv;
        real t = c->g0_c2 * oef;
        real g0 = g0_c1 * (t * t);
        g = (c->half_g2 * g0) * c->td2;
    } else {
        bw = dbv; /* signals.py:110 */
    }
    real tw = R(1) - bw;
    for (int t = 0; t < c->T; ++t) {
        real tissue;
        if (P->full_model) {
            real F = qbo_tissue_F(c->taus[t] * dw);
            tissue = r_exp(-dbv * F) * c->e_te_r2t;
        } else { /* signals.py:194-207 */
            real tc = R(1.0) / dw;
            real r2p = dw * dbv;
            real rt = r2p * c->taus[t];
            if (r_fabs(c->taus[t]) < tc)
                tissue = r_exp(c->r2t_te) * r_exp(-(R(0.3) * (rt * rt)) / dbv);
            else
                tissue = r_exp(c->r2t_te) * r_exp(dbv - rt);
        }
        real blood = 0;
        if (P->include_blood) blood = c->e_r2b_te * r_exp(-g * c->blood_B[t]);
        out[t] = tw * tissue + bw * blood;
    }
}

static inline void signal_one(const qbo_phys *P, const fwd_consts *c, real oef, real dbv,
                              real *out) {
    signal_one_h(P, c, oef, dbv, R(-1), out);
}

/* signals.py:64-96: optional per-voxel haematocrit and the misalignment augmentation with its
 * random draws made explicit: images t > from_idx[v] come from alt[v] = the perturbed (OEF, DBV)
 * (the reference blends with a 0/1 mask, :95-96, which selects exactly). */
void qbo_signal_fwd_ex(const qbo_phys *P, const real *oef_dbv, const real *hct, const real *alt,
                       const int32_t *from_idx, real *signal, int64_t V) {
    fwd_consts c;
    make_consts(P, &c);
    init_nodes();
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < V; ++v) {
        real h = hct ? hct[v] : R(-1);
        real *out = signal + v * c.T;
        signal_one_h(P, &c, oef_dbv[2 * v], oef_dbv[2 * v + 1], h, out);
        if (alt && from_idx && from_idx[v] < c.T - 1) {
            real tmp[QBO_MAX_T];
            signal_one_h(P, &c, alt[2 * v], alt[2 * v + 1], h, tmp);
            for (int t = from_idx[v] < 0 ? 0 : from_idx[v] + 1; t < c.T; ++t) out[t] = tmp[t];
        }
    }
}

void qbo_signal_fwd(const qbo_phys *P, const real *oef_dbv, real *signal, int64_t V) {
    fwd_consts c;
    make_consts(P, &c);
    init_nodes();
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < V; ++v)
        signal_one(P, &c, oef_dbv[2 * v], oef_dbv[2 * v + 1], signal + v * c.T);
}

void qbo_signal_jac(const qbo_phys *P, const real *oef_dbv, real *jac, int64_t V) {
    fwd_consts c;
    make_consts(P, &c);
    init_nodes();
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < V; ++v) {
        real oef = oef_dbv[2 * v], dbv = oef_dbv[2 * v + 1];
        real dw = c.dw_coef * oef;
        real bw = P->include_blood ? c.m_bld_nb * dbv : dbv;
        real dbw = P->include_blood ? c.m_bld_nb : R(1);
        real t2 = c.g0_c2 * oef;
        real g = (c.half_g2 * (c.g0_c1 * (t2 * t2))) * c.td2;
        real dg = oef != 0 ? R(2) * g / oef : 0;
        for (int t = 0; t < c.T; ++t) {
            real x = c.taus[t] * dw;
            real tissue, dt_doef, dt_ddbv;
            if (P->full_model) {
                real F = qbo_tissue_F(x), dF = qbo_tissue_dF(x);
                tissue = r_exp(-dbv * F) * c.e_te_r2t;
                dt_ddbv = -F * tissue;
                dt_doef = -dbv * dF * (c.taus[t] * c.dw_coef) * tissue;
            } else {
                real tc = R(1.0) / dw, r2p = dw * dbv, rt = r2p * c.taus[t];
                if (r_fabs(c.taus[t]) < tc) {
                    tissue = r_exp(c.r2t_te) * r_exp(-(R(0.3) * (rt * rt)) / dbv);
                    /* exponent = -0.3 (dw tau)^2 dbv */
                    real k = c.taus[t] * dw;
                    dt_ddbv = -R(0.3) * k * k * tissue;
                    dt_doef = -R(0.6) * k * (c.taus[t] * c.dw_coef) * dbv * tissue;
                } else {
                    tissue = r_exp(c.r2t_te) * r_exp(dbv - rt);
                    dt_ddbv = (R(1) - dw * c.taus[t]) * tissue;
                    dt_doef = -(c.dw_coef * dbv * c.taus[t]) * tissue;
                }
            }
            real blood = 0, db_doef = 0;
            if (P->include_blood) {
                blood = c.e_r2b_te * r_exp(-g * c.blood_B[t]);
                db_doef = -dg * c.blood_B[t] * blood;
            }
            real *j = jac + (v * c.T + t) * 2;
            j[0] = (R(1) - bw) * dt_doef + bw * db_doef;
            j[1] = (R(1) - bw) * dt_ddbv + dbw * (blood - tissue);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Encoder (voxel-wise) -- model.py:97-113 (normalise_data), :115-223 (create_encoder).
 * ---------------------------------------------------------------------------------------- */
static inline real clipr(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline void normalise_one(const qbo_loss_cfg *C, const real *x, real *n, int T) {
    real c[QBO_MAX_T];
    for (int t = 0; t < T; ++t) c[t] = clipr(x[t], R(1e-2), R(1e8)); /* model.py:101 */
    real den;
    if (C->multi_image_normalisation) /* model.py:104: reduce_mean over se-1..se+1 */
        den = (c[C->se_idx - 1] + c[C->se_idx] + c[C->se_idx + 1]) / R(3);
    else
        den = c[C->se_idx]; /* model.py:106 */
    for (int t = 0; t < T; ++t) n[t] = r_log(c[t] / den); /* model.py:108 */
}

void qbo_normalise(const qbo_loss_cfg *C, const real *x, real *n, int T, int64_t N) {
    for (int64_t i = 0; i < N; ++i) normalise_one(C, x + i * T, n + i * T, T);
}

static inline real sigmoidr(real v) { return R(1) / (R(1) + r_exp(-v)); }

/* ranges of forward_transform -- model.py:88-91 */
#define OEF_RANGE R(0.8)
#define MIN_OEF R(0.04)
#define DBV_RANGE R(0.2)
#define MIN_DBV R(0.001)

/* y[o] = act(sum_i x[i] W[i][o] + b[o]) ; W is [nin][nout] (Keras kernel orientation). */
/* Test hook for the reduced-precision encoder mode (BASELINE config 5, "bf16 forward / fp32 ELBO
 * accum"): both operands of every dense product are rounded to bfloat16 (round-to-nearest-even),
 * products and sums stay in `real`; biases, activations functions and heads stay in `real`. */
static int g_encoder_bf16 = 0;
void qbo_set_encoder_bf16(int on) { g_encoder_bf16 = on; }
static inline real bf16_round(real v) {
    if (!g_encoder_bf16) return v;
    float f = (float)v;
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return v; /* inf / nan */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return (real)f;
}

/* activation_type of EncoderTrainer (model.py:60, 115-120, 151, 155): 'relu', or Keras' 'gelu' =
 * tf.keras.activations.gelu(x, approximate=False) = 0.5 x (1 + erf(x / sqrt 2)).  Process-global test hook. */
static int g_gelu = 0;
void qbo_set_activation_gelu(int on) { g_gelu = on; }
static real activate(real v) {
    if (!g_gelu) return v > 0 ? v : 0;
#ifdef QBO_DOUBLE
    return R(0.5) * v * (R(1.0) + erf(v * R(0.70710678118654752440)));
#else
    return R(0.5) * v * (R(1.0) + erff(v * R(0.70710678118654752440)));
#endif
}

static void dense(const real *x, const real *W, const real *b, real *y, int nin, int nout,
                  int relu) {
    for (int o = 0; o < nout; ++o) {
        real acc = 0;
        for (int i = 0; i < nin; ++i) acc += bf16_round(x[i]) * bf16_round(W[(int64_t)i * nout + o]);
        acc += b[o];
        y[o] = relu ? activate(acc) : acc;
    }
}

void qbo_encoder_fwd(const qbo_weights *W, const qbo_loss_cfg *C, const real *x, real *out1,
                     real *out2, real *sigma, int64_t N) {
    const int T = W->T, U = W->U, L = W->L;
    const int G = W->channelwise_gating ? U : 1;
    const real goff = R(W->gate_offset);
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < N; ++v) {
        real n[QBO_MAX_T];
        real *a = (real *)malloc(sizeof(real) * U * 7);
        real *b = a + U, *skip = b + U, *r = skip + U, *t1 = r + U, *gt = t1 + U, *tmp = gt + U;
        normalise_one(C, x + v * T, n, T);  /* model.py:178 */
        dense(n, W->W0, W->b0, a, T, U, 1); /* model.py:181 */
        memcpy(b, a, sizeof(real) * U);     /* model.py:185: net2 = net1 */
        for (int l = 0; l < L; ++l) {       /* create_block, model.py:142-174 */
            const real *Wc = W->Wc + (int64_t)l * U * U, *bc = W->bc + l * U;
            dense(a, Wc, bc, tmp, U, U, 1); /* stream 1, model.py:145 */
            memcpy(a, tmp, sizeof(real) * U);
            dense(b, Wc, bc, skip, U, U, 1); /* shared conv as skip, model.py:148 */
            for (int i = 0; i < U; ++i) tmp[i] = activate(b[i]); /* model.py:151 */
            /* a (N,1,1,1,T) batch sees only the centre tap of a 'same'-padded 3x3x1 kernel */
            const int64_t ts = (int64_t)(W->taps == 9 ? 9 : 1) * U * U, tc = W->taps == 9 ? 4 * U * U : 0;
            dense(tmp, W->Wr1 + l * ts + tc, W->br1 + l * U, t1, U, U, 1); /* :152,155 */
            dense(t1, W->Wr2 + l * ts + tc, W->br2 + l * U, r, U, U, 0);   /* :156 */
            dense(r, W->Wg + (int64_t)l * U * G, W->bg + l * G, gt, U, G, 0);     /* :164 */
            for (int i = 0; i < U; ++i) { /* gate_convs, model.py:167-170 */
                real gate = sigmoidr(gt[G == 1 ? 0 : i] + goff);
                b[i] = skip[i] * (R(1.0) - gate) + r[i] * gate;
            }
        }
        real o[5];
        if (out1) {
            dense(a, W->Wf, W->bf, o, U, 5, 0); /* model.py:199 */
            memcpy(out1 + v * 5, o, sizeof(o));
        }
        if (out2) {
            dense(b, W->Wf, W->bf, o, U, 5, 0); /* model.py:208 */
            memcpy(out2 + v * 5, o, sizeof(o));
        }
        if (sigma) {
            real s[QBO_MAX_T];
            dense(b, W->Ws, W->bs, s, U, T, 0); /* model.py:211-214,220 */
            for (int t = 0; t < T; ++t) sigma[v * T + t] = r_exp(s[t]);
        }
        free(a);
    }
}

/* ------------------------------------------------------------------------------------------
 * Encoder on image volumes: stream 2 with its 3x3x1 'same' convolutions -- model.py:142-174.
 * Layer by layer over the whole volume (the convolutions couple x/y neighbours).
 * ---------------------------------------------------------------------------------------- */
static void conv3x3(const real *in, const real *K /*[3][3][U][U]*/, const real *b, real *out, int B,
                    int X, int Y, int Z, int U, int relu_out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bb = 0; bb < B; ++bb)
        for (int x = 0; x < X; ++x)
            for (int y = 0; y < Y; ++y)
                for (int z = 0; z < Z; ++z) {
                    real *o = out + ((((int64_t)bb * X + x) * Y + y) * Z + z) * U;
                    for (int c = 0; c < U; ++c) o[c] = 0;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) {
                            const int xx = x + i - 1, yy = y + j - 1;
                            if (xx < 0 || xx >= X || yy < 0 || yy >= Y) continue; /* zero padding */
                            const real *v = in + ((((int64_t)bb * X + xx) * Y + yy) * Z + z) * U;
                            const real *k = K + (int64_t)(i * 3 + j) * U * U;
                            for (int ci = 0; ci < U; ++ci) {
                                const real a = v[ci];
                                for (int c = 0; c < U; ++c) o[c] += a * k[(int64_t)ci * U + c];
                            }
                        }
                    for (int c = 0; c < U; ++c) {
                        o[c] += b[c];
                        if (relu_out) o[c] = activate(o[c]);
                    }
                }
}

/* add_normalizer (model.py:131-140) in front of the residual path's two activations (model.py:150-151, 154-155):
 * keras Dropout (training; the LIBRARY's mask stream, include/qbold_hip.h: element (row, c) of normalizer `layer` is
 * dropped iff half-word c & 7 of Philox4x32-7(ctr = (row_lo, row_hi, (c >> 3) | layer << 16, 5), key = seed) is below
 * rate 2^16), then tfa GroupNormalization(groups = 1, axis = -1): mean and biased variance over all positions and
 * channels of one batch element, epsilon 1e-3, per-channel gamma / beta.  Test hook like the activation's. */
static const real *g_ln = NULL;      /* [L][4][U]: gamma1, beta1, gamma2, beta2 per block */
static double g_drop_rate = 0.0;
static uint64_t g_drop_seed = 0;
void qbo_set_normalizer(const real *ln, double dropout_rate, uint64_t dropout_seed) {
    g_ln = ln;
    g_drop_rate = dropout_rate;
    g_drop_seed = dropout_seed;
}
void qbo_philox4x32_7(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
static real drop_factor(int64_t row, int col, int layer) {
    if (!(g_drop_rate > 0.0) || g_drop_seed == 0) return R(1.0);
    double rate = g_drop_rate < 0.999 ? g_drop_rate : 0.999;
    uint32_t thresh = (uint32_t)lrintf((float)rate * 65536.0f);
    uint32_t ctr[4] = {(uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)(col >> 3) | ((uint32_t)layer << 16), 5u};
    uint32_t key[2] = {(uint32_t)g_drop_seed, (uint32_t)(g_drop_seed >> 32)}, o[4];
    qbo_philox4x32_7(ctr, key, o);
    uint32_t w = o[(col >> 1) & 3];
    uint32_t hw = (col & 1) ? (w >> 16) : (w & 0xffffu);
    return hw < thresh ? R(0.0) : R(1.0 / (1.0 - (double)thresh / 65536.0));
}
/* out = act(LN(Dropout(in))) for a [B][rows_per_b][U] tensor */
static void normalizer(const real *in, real *out, int B, int64_t rows_per_b, int U, int layer, int l, int which) {
    for (int b = 0; b < B; ++b) {
        const int64_t r0 = (int64_t)b * rows_per_b, n = rows_per_b * U;
        double mean = 0, var = 0;
        if (g_ln) {
            for (int64_t i = 0; i < n; ++i) mean += (double)(in[r0 * U + i] * drop_factor(r0 + i / U, (int)(i % U), layer));
            mean /= (double)n;
            for (int64_t i = 0; i < n; ++i) {
                double u = (double)(in[r0 * U + i] * drop_factor(r0 + i / U, (int)(i % U), layer)) - mean;
                var += u * u;
            }
            var /= (double)n;
        }
        const real rstd = R(1.0 / sqrt(var + 1e-3));
        for (int64_t i = 0; i < n; ++i) {
            const int c = (int)(i % U);
            real v = in[r0 * U + i] * drop_factor(r0 + i / U, c, layer);
            if (g_ln) {
                const real *gb = g_ln + ((int64_t)l * 4 + 2 * which) * U;
                v = (v - R(mean)) * rstd * gb[c] + gb[U + c];
            }
            out[r0 * U + i] = activate(v);
        }
    }
}

void qbo_encoder_fwd_spatial(const qbo_weights *W, const qbo_loss_cfg *C, const real *x, int B, int X,
                             int Y, int Z, real *out2, real *sigma) {
    const int T = W->T, U = W->U, L = W->L;
    const int G = W->channelwise_gating ? U : 1;
    const real goff = R(W->gate_offset);
    const int64_t V = (int64_t)B * X * Y * Z;
    real *b = (real *)malloc(sizeof(real) * V * U * 4);
    real *skip = b + V * U, *t1 = skip + V * U, *r = t1 + V * U;
    for (int64_t v = 0; v < V; ++v) {
        real n[QBO_MAX_T];
        normalise_one(C, x + v * T, n, T);
        dense(n, W->W0, W->b0, b + v * U, T, U, 1);
    }
    for (int l = 0; l < L; ++l) {
        const real *Wc = W->Wc + (int64_t)l * U * U, *bc = W->bc + l * U;
        for (int64_t v = 0; v < V; ++v) dense(b + v * U, Wc, bc, skip + v * U, U, U, 1); /* :148 */
        normalizer(b, r, B, (int64_t)X * Y * Z, U, 2 * l, l, 0);                          /* :150-151 */
        conv3x3(r, W->Wr1 + (int64_t)l * 9 * U * U, W->br1 + l * U, t1, B, X, Y, Z, U, 0); /* :152 */
        normalizer(t1, t1, B, (int64_t)X * Y * Z, U, 2 * l + 1, l, 1);                    /* :154-155 */
        conv3x3(t1, W->Wr2 + (int64_t)l * 9 * U * U, W->br2 + l * U, r, B, X, Y, Z, U, 0); /* :156 */
        for (int64_t v = 0; v < V; ++v) {
            real gt[256];
            dense(r + v * U, W->Wg + (int64_t)l * U * G, W->bg + l * G, gt, U, G, 0);      /* :164 */
            for (int i = 0; i < U; ++i) {
                real gate = sigmoidr(gt[G == 1 ? 0 : i] + goff);
                b[v * U + i] = skip[v * U + i] * (R(1.0) - gate) + r[v * U + i] * gate;    /* :170 */
            }
        }
    }
    for (int64_t v = 0; v < V; ++v) {
        if (out2) dense(b + v * U, W->Wf, W->bf, out2 + v * 5, U, 5, 0);
        if (sigma) {
            real s[QBO_MAX_T];
            dense(b + v * U, W->Ws, W->bs, s, U, T, 0);
            for (int t = 0; t < T; ++t) sigma[v * T + t] = r_exp(s[t]);
        }
    }
    free(b);
}

/* smoothness_loss numerator -- model.py:735-752 */
double qbo_smoothness_sum(const real *q, const real *mask, int B, int X, int Y, int Z) {
    double acc = 0;
    for (int bb = 0; bb < B; ++bb)
        for (int x = 0; x < X; ++x)
            for (int y = 0; y < Y; ++y)
                for (int z = 0; z < Z; ++z) {
                    const int64_t v = (((int64_t)bb * X + x) * Y + y) * Z + z;
                    /* forward_transform / range: sigmoid + min/range */
                    const real po = sigmoidr(q[5 * v]) + MIN_OEF / OEF_RANGE;
                    const real pd = sigmoidr(q[5 * v + 2]) + MIN_DBV / DBV_RANGE;
                    if (x + 1 < X) {
                        const int64_t w = v + (int64_t)Y * Z;
                        if (mask[v] > 0 && mask[w] > 0)
                            acc += r_fabs(po - (sigmoidr(q[5 * w]) + MIN_OEF / OEF_RANGE)) +
                                   r_fabs(pd - (sigmoidr(q[5 * w + 2]) + MIN_DBV / DBV_RANGE));
                    }
                    if (y + 1 < Y) {
                        const int64_t w = v + Z;
                        if (mask[v] > 0 && mask[w] > 0)
                            acc += r_fabs(po - (sigmoidr(q[5 * w]) + MIN_OEF / OEF_RANGE)) +
                                   r_fabs(pd - (sigmoidr(q[5 * w + 2]) + MIN_DBV / DBV_RANGE));
                    }
                }
    return acc;
}

/* ------------------------------------------------------------------------------------------
 * Logit-Normal pieces -- model.py:288-316 = logit_mvn.py:72-100, :376-447 = logit_mvn.py:20-70.
 * ---------------------------------------------------------------------------------------- */
static inline real transform_std(real p) { return (r_tanh(p) * R(3.0)) - R(1.0); } /* :288-290 */
static inline real transform_offdiag(real p) { return r_tanh(p) * R(exp(-2.0)); }  /* :292-294 */


static inline void reparam_one(const real *q, real z0, real z1, real *oef, real *dbv) {
    /* model.py:25-31 */
    real a = q[0] + z0 * r_exp(transform_std(q[1]));
    real b = q[2] + z0 * transform_offdiag(q[4]) + z1 * r_exp(transform_std(q[3]));
    /* forward_transform, model.py:299-305 */
    *oef = (sigmoidr(a) * OEF_RANGE) + MIN_OEF;
    *dbv = (sigmoidr(b) * DBV_RANGE) + MIN_DBV;
}

void qbo_reparam(const real *q, const real *z, real *oef_dbv, int64_t N) {
    for (int64_t i = 0; i < N; ++i)
        reparam_one(q + 5 * i, z[2 * i], z[2 * i + 1], oef_dbv + 2 * i, oef_dbv + 2 * i + 1);
}

static inline real nlogp_one(real oef, real dbv, const real *p) {
    real oef_mean = p[0], dbv_mean = p[2];
    real s_o = transform_std(p[1]), s_d = transform_std(p[3]); /* model.py:381,383 */
    real cov = transform_offdiag(p[4]);                        /* model.py:397 */
    /* backwards_transform(include_logit=False), model.py:307-311 */
    real x0 = (oef - MIN_OEF) / OEF_RANGE;
    real x1 = (dbv - MIN_DBV) / DBV_RANGE;
    const real eps = R(1e-6);
    x0 = clipr(x0, eps, R(1.0) - eps); /* model.py:395 */
    x1 = clipr(x1, eps, R(1.0) - eps);
    real l0 = r_log(x0 / (R(1.0) - x0)); /* logit, model.py:10-12 */
    real l1 = r_log(x1 / (R(1.0) - x1));
    /* squared_whitened_residual, model.py:423-441 */
    real inv_tl = r_exp(s_o * R(-1.0));
    real inv_br = r_exp(s_d * R(-1.0));
    real inv_bl = r_exp(s_o * R(-1.0) + s_d * R(-1.0)) * cov * R(-1.0);
    real r0 = l0 - oef_mean, r1 = l1 - dbv_mean;
    real w0 = r0 * inv_tl;
    real w1 = r1 * inv_br + r0 * inv_bl;
    real swr = w0 * w0 + w1 * w1;
    real log_det = R(2.0) * (s_o + s_d); /* model.py:443-447 */
    /* gaussian_nll_chol, model.py:385-390 */
    real loss = -(-r_log(R(2.0 * M_PI)) - R(0.5) * log_det - R(0.5) * swr);
    /* Jacobian, model.py:398 */
    loss = loss + ((r_log(x0) + r_log(R(1.0) - x0)) + (r_log(x1) + r_log(R(1.0) - x1)));
    return loss;
}

void qbo_logit_mvn_nlogp(const real *y, const real *p, real *out, int64_t N) {
    for (int64_t i = 0; i < N; ++i) out[i] = nlogp_one(y[2 * i], y[2 * i + 1], p + 5 * i);
}

/* synthetic_data_loss, use_mvg, use_r2p_loss=False, inv_gamma off -- model.py:449-471,514 */
double qbo_synthetic_data_loss(const real *y_true, const real *q, int64_t N) {
    double acc = 0;
    for (int64_t i = 0; i < N; ++i) acc += nlogp_one(y_true[3 * i], y_true[3 * i + 1], q + 5 * i);
    return acc / (double)N;
}

/* ... with the inverse-gamma prior on the marginal variances (model.py:492-507, use_mvg branch, fixed
 * alpha / beta): oef_var = exp(oef_log_std)^2, dbv_var = exp(dbv_log_std)^2 + y_pred[:,4]^2 (the raw
 * fifth parameter, :499); loss = nlogp - IG(a,b).log_prob(oef_var) - IG(a,b).log_prob(dbv_var) with
 * tfp InverseGamma.log_prob(x) = a log b - lgamma(a) - (a + 1) log x - b / x. */
double qbo_synthetic_data_loss_ig(const real *y_true, const real *q, double alpha, double beta,
                                  int64_t N) {
    if (!(alpha * beta > 0.0)) return qbo_synthetic_data_loss(y_true, q, N);
    double acc = 0;
    const real a = R(alpha), b = R(beta);
    const real c0 = R(alpha * log(beta) - lgamma(alpha));
    for (int64_t i = 0; i < N; ++i) {
        const real *p = q + 5 * i;
        real so = R(3) * r_tanh(p[1]) - R(1), sd = R(3) * r_tanh(p[3]) - R(1); /* transform_std */
        real eo = r_exp(so), ed = r_exp(sd);
        real xo = eo * eo, xd = ed * ed + p[4] * p[4];
        real lp = (c0 - (a + R(1)) * r_log(xo) - b / xo) + (c0 - (a + R(1)) * r_log(xd) - b / xd);
        acc += nlogp_one(y_true[3 * i], y_true[3 * i + 1], p) - lp;
    }
    return acc / (double)N;
}

/* ------------------------------------------------------------------------------------------
 * NLL -- fine_tune_loss_fn, model.py:527-568 (per voxel, before the mask multiply :564).
 * ---------------------------------------------------------------------------------------- */
static inline real nll_one(const qbo_loss_cfg *C, const real *x, real mask, const real *pred,
                           const real *sigma, int T, real t_const) {
    real nt, np_;
    const int se = C->se_idx;
    if (C->multi_image_normalisation) { /* model.py:541-542 */
        nt = (x[se - 1] + x[se] + x[se + 1]) / R(3) + R(1e-3);
        np_ = (pred[se - 1] + pred[se] + pred[se + 1]) / R(3) + R(1e-3);
    } else { /* model.py:544-545 */
        nt = x[se] + R(1e-3);
        np_ = pred[se] + R(1e-3);
    }
    real acc = 0;
    for (int t = 0; t < T; ++t) {
        real yt = x[t] / nt, yp = pred[t] / np_;
        if (C->predict_log_data) { /* model.py:547-549 */
            yt = mask > 0 ? r_log(yt) : 0;
            yp = mask > 0 ? r_log(yp) : 0;
        }
        real res = yt - yp; /* model.py:552 */
        real s = sigma[t];
        real nll;
        if (C->use_student_t) { /* tfp StudentT(df,0,s).log_prob, model.py:557-559 */
            real df = R(C->student_t_df);
            real yy = res / s;
            nll = -(t_const - r_log(s) - R(0.5) * (df + R(1)) * r_log1p(yy * yy / df));
        } else { /* model.py:561 */
            real rs = res / s;
            nll = -(-r_log(s) - R(log(sqrt(2.0 * M_PI))) - R(0.5) * (rs * rs));
        }
        acc += nll; /* model.py:563 */
    }
    return acc;
}

static real student_t_const(const qbo_loss_cfg *C) {
    if (!C->use_student_t) return 0;
    double df = C->student_t_df;
    return R(lgamma(0.5 * (df + 1.0)) - lgamma(0.5 * df) - 0.5 * log(df) - 0.5 * log(M_PI));
}

void qbo_nll(const qbo_loss_cfg *C, const real *x, const real *mask, const real *pred,
             const real *sigma, real *nll, int T, int64_t N) {
    real tc = student_t_const(C);
    for (int64_t i = 0; i < N; ++i)
        nll[i] = nll_one(C, x + i * T, mask ? mask[i] : R(1), pred + i * T, sigma + i * T, T, tc);
}

/* ------------------------------------------------------------------------------------------
 * KL -- mvg_kl_samples model.py:592-610, closed form mvg_kl model.py:612-652.
 * ---------------------------------------------------------------------------------------- */
static inline real kl_samples_one(const real *q, const real *prior, const real *z, int K) {
    real acc = 0;
    for (int k = 0; k < K; ++k) {
        real oef, dbv;
        reparam_one(q, z[2 * k], z[2 * k + 1], &oef, &dbv); /* create_samples, model.py:318-324 */
        real log_q = -nlogp_one(oef, dbv, q);               /* model.py:596 */
        real log_p = -nlogp_one(oef, dbv, prior);           /* model.py:597 */
        acc += log_q - log_p;                               /* model.py:603 */
    }
    return acc / R(K); /* model.py:609 */
}

void qbo_kl_samples(const real *q, const real *prior, const real *z, int K, real *kl, int64_t N) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i)
        kl[i] = kl_samples_one(q + 5 * i, prior + 5 * i, z + (int64_t)i * K * 2, K);
}

void qbo_kl_closed(const real *q, const real *prior, real *kl, int64_t N) {
    for (int64_t i = 0; i < N; ++i) {
        const real *qq = q + 5 * i, *pp = prior + 5 * i;
        real q_cov = transform_offdiag(qq[4]), p_cov = transform_offdiag(pp[4]); /* :622-623 */
        real q_so = transform_std(qq[1]), q_sd = transform_std(qq[3]);
        real p_so = transform_std(pp[1]), p_sd = transform_std(pp[3]);
        real det_q = R(2.0) * (q_so + q_sd), det_p = R(2.0) * (p_so + p_sd); /* :628-629 */
        /* squared_whitened_residual(p_mu, q_mu, p_log_stds, p_cov) :633 */
        real inv_tl = r_exp(-p_so), inv_br = r_exp(-p_sd);
        real inv_bl = r_exp(-p_so - p_sd) * p_cov * R(-1.0);
        real r0 = pp[0] - qq[0], r1 = pp[2] - qq[2];
        real w0 = r0 * inv_tl, w1 = r1 * inv_br + r0 * inv_bl;
        real sq = w0 * w0 + w1 * w1;
        real det_term = det_p - det_q; /* :635 */
        real inv_p_tl = R(1.0) / r_exp(p_so), inv_p_br = R(1.0) / r_exp(p_sd); /* :637-638 */
        real inv_p_od = inv_p_tl * p_cov * inv_p_br * R(-1.0);
        real inv_pcov_tl = inv_p_tl * inv_p_tl;
        real inv_pcov_br = inv_p_od * inv_p_od + inv_p_br * inv_p_br;
        real inv_pcov_od = inv_p_tl * inv_p_od;
        real q_tl = r_exp(q_so) * r_exp(q_so);
        real q_br = r_exp(q_sd) * r_exp(q_sd) + q_cov * q_cov;
        real q_od = q_cov * r_exp(q_so);
        real trace = inv_pcov_tl * q_tl + inv_pcov_od * q_od + inv_pcov_od * q_od +
                     q_br * inv_pcov_br; /* :648 */
        kl[i] = R(0.5) * (trace + sq + det_term - R(2.0)); /* :651 */
    }
}

/* kl_loss, diagonal family (use_mvg = False, no population prior) -- model.py:686-716:
 * tfp LogitNormal(loc, scale=exp(log_std)).kl_divergence = tfp kl_normal_normal of the base Normals:
 *   b_inv = 1 / b.scale;  d = log(a.scale) - log(b.scale);
 *   0.5 * squared_difference(a.loc * b_inv, b.loc * b_inv) + 0.5 * expm1(2 d) - d
 * with log_std = transform_std(raw) (:704-707).  q, prior rows 5 wide, columns 0-3 used. */
void qbo_kl_diag(const real *q, const real *prior, real *kl, int64_t N) {
    for (int64_t i = 0; i < N; ++i) {
        const real *qq = q + 5 * i, *pp = prior + 5 * i;
        real acc = 0;
        for (int dim = 0; dim < 2; ++dim) {
            real a_loc = qq[2 * dim], b_loc = pp[2 * dim];
            real a_scale = r_exp(transform_std(qq[2 * dim + 1]));
            real b_scale = r_exp(transform_std(pp[2 * dim + 1]));
            real b_inv = R(1.0) / b_scale;
            real d = r_log(a_scale) - r_log(b_scale);
            real sd = a_loc * b_inv - b_loc * b_inv;
#ifdef QBO_DOUBLE
            real em = expm1(R(2.0) * d);
#else
            real em = expm1f(R(2.0) * d);
#endif
            acc += R(0.5) * (sd * sd) + R(0.5) * em - d;
        }
        kl[i] = acc;
    }
}

/* kl_loss, diagonal family WITH the population prior (use_mvg = False, use_population_prior = True,
 * mog_components = 1) -- model.py:687-690, 704-716: 'predictions' carries [q4 | prior4]; the per-voxel
 * "true" tensor contributes its mask only.  Per voxel the KL is qbo_kl_diag against the one population
 * prior.  The returned value is the batch cost that joins the KL numerator (:710-713, :721):
 *   ig = InverseGamma(1, 2);  cost = -ig.log_prob(exp(2 mean(p_dbv_log_std))) - ig.log_prob(exp(2 mean(
 *   p_oef_log_std)));  cost *= shape(predicted)[0]       (the BATCH axis, not the voxel count)
 * with log_std = transform_std(raw); tfp InverseGamma.log_prob(x) = c log s - lgamma(c) - (c + 1) log x - s / x. */
double qbo_population_prior_cost(const real *prior4 /*[4]*/, int batch) {
    double cost = 0;
    const int order[2] = {3, 1}; /* DBV first, then OEF (:711-712) */
    for (int k = 0; k < 2; ++k) {
        double log_std = (double)transform_std(prior4[order[k]]);
        double v = exp(2.0 * log_std);
        cost -= 1.0 * log(2.0) - lgamma(1.0) - (1.0 + 1.0) * log(v) - 2.0 / v;
    }
    return cost * (double)batch;
}

/* kl_loss, mixture-of-Gaussians population prior (use_mvg = False, use_population_prior = True, mog_components = M > 1)
 * -- model.py:666-685.  Per voxel, with ONE reparameterised draw per dimension (z[0] for OEF, z[1] for DBV, :672-675):
 *   entropy = s_o + s_d;  sample = mu + z exp(s);
 *   kl = -entropy + (1 / M) sum_i [ nll(oef_sample; comp_i[0], comp_i[1]) + nll(dbv_sample; comp_i[2], comp_i[3]) ],
 *   nll(x; m, raw) = transform_std(raw) + 0.5 ((x - m) / exp(transform_std(raw)))^2                       (:677-678)
 * q rows 5 wide (columns 0-3 used), comps [M][4] raw parameters. */
void qbo_kl_mog(const real *q, const real *comps, int M, const real *z /*[N][2]*/, real *kl, int64_t N) {
    for (int64_t i = 0; i < N; ++i) {
        const real *qq = q + 5 * i;
        const real so = transform_std(qq[1]), sd = transform_std(qq[3]);
        const real xo = qq[0] + z[2 * i] * r_exp(so), xd = qq[2] + z[2 * i + 1] * r_exp(sd);
        real acc = -(so + sd);
        for (int c = 0; c < M; ++c) {
            const real *p = comps + 4 * c;
            const real po = transform_std(p[1]), pd = transform_std(p[3]);
            const real ro = (xo - p[0]) / r_exp(po), rd = (xd - p[2]) / r_exp(pd);
            acc += (po + R(0.5) * ro * ro) / (real)M;
            acc += (pd + R(0.5) * rd * rd) / (real)M;
        }
        kl[i] = acc;
    }
}

/* logit_gaussian_log_prob (diagonal family, model.py:406-421): NEGATIVE log-density up to the
 * reference's own constant -- gaussian_nll (:403-404) carries no log sqrt(2 pi).  p rows 5 wide. */
void qbo_logit_gaussian_nlogp(const real *y, const real *p, real *out, int64_t N) {
    for (int64_t i = 0; i < N; ++i) {
        const real *pp = p + 5 * i;
        real x0 = (y[2 * i] - R(0.04)) / R(0.8), x1 = (y[2 * i + 1] - R(0.001)) / R(0.2); /* backwards_transform */
        real l0 = r_log(x0) - r_log(R(1.0) - x0), l1 = r_log(x1) - r_log(R(1.0) - x1);     /* logit */
        real so = transform_std(pp[1]), sd = transform_std(pp[3]);
        real r0 = (l0 - pp[0]) / r_exp(so), r1 = (l1 - pp[2]) / r_exp(sd);
        real lo = so + R(0.5) * (r0 * r0), ld = sd + R(0.5) * (r1 * r1);
        out[i] = lo + ld + (r_log(x0 * (R(1.0) - x0)) + r_log(x1 * (R(1.0) - x1)));
    }
}

/* ------------------------------------------------------------------------------------------
 * Posterior moments -- calculate_means(include_r2p=True, return_stds=True), model.py:326-343.
 * ---------------------------------------------------------------------------------------- */
void qbo_moments(const qbo_phys *P, const real *q, const real *z, int n, real *means, real *vars,
                 int64_t N) {
    real dw_coef = R((4.0 / 3.0) * M_PI * P->gamma * P->b0 * P->dchi * P->hct);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        real *so = (real *)malloc(sizeof(real) * 3 * n);
        real *sd = so + n, *sr = sd + n;
        real mo = 0, md = 0, mr = 0;
        for (int k = 0; k < n; ++k) {
            const real *zz = z + ((int64_t)i * n + k) * 2;
            reparam_one(q + 5 * i, zz[0], zz[1], &so[k], &sd[k]);
            sr[k] = (dw_coef * so[k]) * sd[k]; /* calculate_r2p, model.py:524-525 */
            mo += so[k];
            md += sd[k];
            mr += sr[k];
        }
        mo /= R(n);
        md /= R(n);
        mr /= R(n);
        real vo = 0, vd = 0, vr = 0;
        for (int k = 0; k < n; ++k) { /* biased variance, model.py:331,337 */
            vo += (so[k] - mo) * (so[k] - mo);
            vd += (sd[k] - md) * (sd[k] - md);
            vr += (sr[k] - mr) * (sr[k] - mr);
        }
        means[3 * i] = mo;
        means[3 * i + 1] = md;
        means[3 * i + 2] = mr;
        if (vars) {
            vars[3 * i] = vo / R(n);
            vars[3 * i + 1] = vd / R(n);
            vars[3 * i + 2] = vr / R(n);
        }
        free(so);
    }
}

/* ------------------------------------------------------------------------------------------
 * One voxel-ELBO evaluation: S reparam samples -> forward model -> NLL,
 * K-sample MC KL, masked sums.  build_fine_tuner model.py:239-286 (S copies :245-246),
 * fine_tune_loss_fn :527-568 (tile :529), kl_loss :654-665, ELBO = nll + kl train.py:351.
 * ---------------------------------------------------------------------------------------- */
void qbo_elbo(const qbo_phys *P, const qbo_loss_cfg *C, const real *x, const real *mask,
              const real *q, const real *prior, const real *sigma, const real *zs, int S,
              const real *zk, int K, real *nll_v, real *kl_v, double *sums, int64_t N) {
    fwd_consts c;
    make_consts(P, &c);
    init_nodes();
    const int T = c.T;
    real tc = student_t_const(C);
    double s_nll = 0, s_kl = 0, s_m = 0;
#pragma omp parallel for schedule(static) reduction(+ : s_nll, s_kl, s_m)
    for (int64_t v = 0; v < N; ++v) {
        real m = mask ? mask[v] : R(1);
        real acc = 0;
        real pred[QBO_MAX_T];
        for (int s = 0; s < S; ++s) {
            real oef, dbv;
            const real *zz = zs + ((int64_t)v * S + s) * 2;
            reparam_one(q + 5 * v, zz[0], zz[1], &oef, &dbv);
            signal_one(P, &c, oef, dbv, pred);
            acc += nll_one(C, x + v * T, m, pred, sigma + v * T, T, tc);
        }
        real nll = acc / R(S);
        real kl = K > 0 ? kl_samples_one(q + 5 * v, prior + 5 * v, zk + (int64_t)v * K * 2, K) : 0;
        if (nll_v) nll_v[v] = nll;
        if (kl_v) kl_v[v] = kl;
        s_nll += (double)(nll * m);        /* model.py:564 */
        s_kl += (double)(m > 0 ? kl : 0);  /* model.py:661 */
        s_m += (double)m;
    }
    sums[0] = s_nll;
    sums[1] = s_kl;
    sums[2] = s_m;
}

/* ------------------------------------------------------------------------------------------
 * Random123 Philox4x32 (Salmon et al., SC'11) and the normal stream shared with the kernels.
 *
 * The stream is the LIBRARY's definition, not the reference's (tf.random.normal cannot be reproduced); this
 * is its restatement, word for word (qbold_vi_amd/csrc/qbold_dev.h).  Round 4:
 *   draw i of (seed, voxel, stream) = word (i & 3) of Philox4x32-7(ctr = (voxel_lo, voxel_hi, i >> 2, stream), key = seed)
 *   word w: u1 = ((w & 0xffff) + 0.5) 2^-16, r = sqrt(-2 ln u1); theta = (w >> 9) 2^-23 revolutions;
 *           (z0, z1) = r (cos 2 pi theta, sin 2 pi theta), everything in double, rounded once.
 * ---------------------------------------------------------------------------------------- */
static void philox4x32_rounds(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void qbo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_rounds(ctr, key, 10, out);
}
void qbo_philox4x32_7(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_rounds(ctr, key, 7, out);
}

static inline void box_muller16(uint32_t w, real *z0, real *z1) {
    double u1 = ((double)(w & 0xffffu) + 0.5) * 0x1p-16;   /* in (0, 1): |z| <= sqrt(-2 ln 2^-17) = 4.8549 */
    double th = (double)(w >> 9) * 0x1p-23;                 /* revolutions: the high half's sixteen bits lead */
    double rr = sqrt(-2.0 * log(u1));
    *z0 = R(rr * cos(2.0 * M_PI * th));
    *z1 = R(rr * sin(2.0 * M_PI * th));
}

void qbo_philox_normals(uint64_t seed, uint32_t stream, int64_t voxel0, int64_t N, int n,
                        real *z) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        uint64_t vox = (uint64_t)(voxel0 + i);
        for (int g = 0; 4 * g < n; ++g) {
            uint32_t ctr[4] = {(uint32_t)vox, (uint32_t)(vox >> 32), (uint32_t)g, stream};
            uint32_t o[4];
            qbo_philox4x32_7(ctr, key, o);
            for (int d = 0; d < 4 && 4 * g + d < n; ++d) {
                real *zz = z + ((int64_t)i * n + 4 * g + d) * 2;
                box_muller16(o[d], &zz[0], &zz[1]);
            }
        }
    }
}

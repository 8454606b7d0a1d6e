"""ctypes front-end of the CPU oracle (oracle/qbold_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/qbold_oracle.h).  Importable only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package
(qbold_vi_amd/) never imports this module.

Two builds of the same C source are exposed: ``Oracle('f32')`` (the parity target: reference
float32 semantics) and ``Oracle('f64')`` (double "truth" used for error budgets).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
QBO_MAX_T = 64


def build(force=False):
    """Compile both oracle builds with gcc (a few seconds)."""
    libs = [os.path.join(_BUILD, f"libqbold_oracle_{p}.so") for p in ("f32", "f64")]
    src = [os.path.join(_HERE, f) for f in ("qbold_oracle.c", "qbold_oracle.h", "Makefile")]
    stale = force or any(
        (not os.path.exists(l)) or os.path.getmtime(l) < max(os.path.getmtime(s) for s in src)
        for l in libs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return libs


class _Phys(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("gamma", "b0", "dchi", "te", "r2t", "tr", "ti", "t1b", "hct",
                 "tau_start", "tau_end", "tau_step")] + \
               [("full_model", C.c_int32), ("include_blood", C.c_int32)]


class _LossCfg(C.Structure):
    _fields_ = [("se_idx", C.c_int32), ("multi_image_normalisation", C.c_int32),
                ("predict_log_data", C.c_int32), ("use_student_t", C.c_int32),
                ("student_t_df", C.c_double)]


def _weights_struct(ptr_t):
    class _Weights(C.Structure):
        _fields_ = [("T", C.c_int32), ("U", C.c_int32), ("L", C.c_int32),
                    ("channelwise_gating", C.c_int32), ("taps", C.c_int32),
                    ("gate_offset", C.c_double)] + \
                   [(n, ptr_t) for n in ("W0", "b0", "Wc", "bc", "Wr1", "br1", "Wr2", "br2",
                                         "Wg", "bg", "Wf", "bf", "Ws", "bs")]
    return _Weights


WEIGHT_NAMES = ("W0", "b0", "Wc", "bc", "Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Wf", "bf",
                "Ws", "bs")


def weight_shapes(T, U, L, channelwise_gating=True):
    G = U if channelwise_gating else 1
    return dict(W0=(T, U), b0=(U,), Wc=(L, U, U), bc=(L, U), Wr1=(L, U, U), br1=(L, U),
                Wr2=(L, U, U), br2=(L, U), Wg=(L, U, G), bg=(L, G), Wf=(U, 5), bf=(5,),
                Ws=(U, T), bs=(T,))


def init_weights(T=11, U=60, L=2, channelwise_gating=True, resid_init_std=0.05,
                 im_loss_sigma=0.05, seed=1, taps=1):
    """Reference initialisers (model.py:119,129,211-214): HeNormal (truncated normal, stddev
    sqrt(2/fan_in)/0.8796) for the 1x1x1 layers, N(0, resid_init_std) for residual / gating /
    sigma-head kernels, zero biases except the sigma head (log im_loss_sigma)."""
    rng = np.random.default_rng(seed)
    shapes = weight_shapes(T, U, L, channelwise_gating)
    if taps == 9:  # full 3x3x1 kernels [L][3][3][U][U] (Keras layout) for the residual convolutions
        shapes["Wr1"] = shapes["Wr2"] = (L, 3, 3, U, U)

    def he(shape, fan_in):
        std = np.sqrt(2.0 / fan_in) / 0.87962566103423978
        out = rng.standard_normal(shape)
        bad = np.abs(out) > 2.0
        while bad.any():
            out[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(out) > 2.0
        return (out * std).astype(np.float32)

    w = {}
    w["W0"] = he(shapes["W0"], T)
    w["Wc"] = he(shapes["Wc"], U)
    w["Wf"] = he(shapes["Wf"], U)
    for n in ("Wr1", "Wr2", "Wg", "Ws"):
        w[n] = (rng.standard_normal(shapes[n]) * resid_init_std).astype(np.float32)
    for n in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[n] = np.zeros(shapes[n], np.float32)
    w["bs"] = np.full(shapes["bs"], np.log(im_loss_sigma), np.float32)
    w["meta"] = dict(T=T, U=U, L=L, channelwise_gating=bool(channelwise_gating), taps=taps)
    return w


DEFAULT_PARAMS = dict(tr="3.0", ti="1.21", te="0.074", tau_start="-0.016", tau_end="0.065",
                      tau_step="0.008", dchi="2.64e-7", gamma="2.67513e8", b0="3.0", t1b="1.58",
                      r2t="11.5", hct="0.34")


class Oracle:
    def __init__(self, precision="f32", params=None, full_model=True, include_blood=True,
                 multi_image_normalisation=False, predict_log_data=False, student_t_df=None,
                 threads=None, node0_zero=False):
        build()
        assert precision in ("f32", "f64")
        self.lib = C.CDLL(os.path.join(_BUILD, f"libqbold_oracle_{precision}.so"))
        self.dtype = np.float32 if precision == "f32" else np.float64
        self.creal = C.c_float if precision == "f32" else C.c_double
        self.preal = C.POINTER(self.creal)
        self._W = _weights_struct(self.preal)
        assert self.lib.qbo_real_bytes() == np.dtype(self.dtype).itemsize
        p = dict(DEFAULT_PARAMS)
        if params is not None:
            p.update({k: params[k] for k in DEFAULT_PARAMS if k in params})
        self.params = p
        self.phys = _Phys(**{k: float(p[k]) for k in DEFAULT_PARAMS},
                          full_model=int(bool(full_model)), include_blood=int(bool(include_blood)))
        se_idx = int(abs(float(p["tau_start"]) / float(p["tau_step"])))  # model.py:95
        use_t = student_t_df is not None and student_t_df < 50  # model.py:557
        self.cfg = _LossCfg(se_idx, int(bool(multi_image_normalisation)),
                            int(bool(predict_log_data)), int(use_t),
                            float(student_t_df) if use_t else 0.0)
        L = self.lib
        L.qbo_j0.restype = self.creal
        L.qbo_j0.argtypes = [self.creal]
        L.qbo_j1.restype = self.creal
        L.qbo_j1.argtypes = [self.creal]
        L.qbo_tissue_F.restype = self.creal
        L.qbo_tissue_F.argtypes = [self.creal]
        L.qbo_tissue_dF.restype = self.creal
        L.qbo_tissue_dF.argtypes = [self.creal]
        L.qbo_synthetic_data_loss.restype = C.c_double
        L.qbo_synthetic_data_loss_ig.restype = C.c_double
        if threads:
            L.qbo_set_threads(int(threads))
        # NB process-global in the C library: use one policy per precision within a test session
        L.qbo_set_node0_zero(int(bool(node0_zero)))
        t = np.zeros(QBO_MAX_T, self.dtype)
        self.T = L.qbo_taus(C.byref(self.phys), self._p(t))
        self.taus = t[:self.T].copy()
        self.se_idx = se_idx

    # -- helpers ------------------------------------------------------------------------
    def _a(self, x, shape=None):
        a = np.ascontiguousarray(x, dtype=self.dtype)
        if shape is not None:
            a = a.reshape(shape)
        return a

    def _p(self, a):
        return a.ctypes.data_as(self.preal)

    def set_encoder_bf16(self, on):
        """Process-global test hook: bf16-rounded operands in the voxel-wise encoder's products."""
        self.lib.qbo_set_encoder_bf16(int(bool(on)))

    def set_activation(self, name):
        """Process-global test hook: 'relu' (default) or 'gelu' in the encoder restatements."""
        assert name in ("relu", "gelu")
        self.lib.qbo_set_activation_gelu(int(name == "gelu"))

    def set_threads(self, n):
        self.lib.qbo_set_threads(int(n))

    def _weights(self, w):
        meta = w["meta"]
        keep = [self._a(w[n]) for n in WEIGHT_NAMES]
        ws = self._W(meta["T"], meta["U"], meta["L"], int(meta["channelwise_gating"]),
                     int(meta.get("taps", 1)), float(w.get("gate_offset", 0.0)),
                     *[self._p(a) for a in keep])
        return ws, keep

    # -- scalar functions ----------------------------------------------------------------
    def j0(self, x):
        x = self._a(x)
        y = np.empty_like(x)
        self.lib.qbo_j0_array(self._p(x), self._p(y), C.c_int64(x.size))
        return y

    def tissue_F(self, x):
        x = self._a(x)
        y = np.empty_like(x)
        self.lib.qbo_tissue_F_array(self._p(x), self._p(y), C.c_int64(x.size))
        return y

    def tissue_dF(self, x):
        x = self._a(x)
        return np.array([self.lib.qbo_tissue_dF(self.creal(v)) for v in x.ravel()],
                        self.dtype).reshape(x.shape)

    # -- forward model -------------------------------------------------------------------
    def signal_fwd(self, oef_dbv):
        x = self._a(oef_dbv)
        assert x.shape[-1] == 2
        V = x.size // 2
        out = np.empty((V, self.T), self.dtype)
        self.lib.qbo_signal_fwd(C.byref(self.phys), self._p(x), self._p(out), C.c_int64(V))
        return out.reshape(x.shape[:-1] + (self.T,))

    def signal_jac(self, oef_dbv):
        x = self._a(oef_dbv)
        V = x.size // 2
        out = np.empty((V, self.T, 2), self.dtype)
        self.lib.qbo_signal_jac(C.byref(self.phys), self._p(x), self._p(out), C.c_int64(V))
        return out

    # -- encoder -------------------------------------------------------------------------
    def normalise(self, x):
        x = self._a(x)
        T = x.shape[-1]
        out = np.empty_like(x)
        self.lib.qbo_normalise(C.byref(self.cfg), self._p(x), self._p(out), T,
                               C.c_int64(x.size // T))
        return out

    def signal_fwd_ex(self, oef_dbv, hct=None, alt=None, from_idx=None):
        """signals.py:64-96 options: per-voxel hct, misalignment (alt [V,2], from_idx [V] int32)."""
        y = self._a(oef_dbv, (-1, 2))
        V = y.shape[0]
        h = None if hct is None else self._a(hct, (V,))
        a = None if alt is None else self._a(alt, (V, 2))
        f = None if from_idx is None else np.ascontiguousarray(from_idx, np.int32).reshape(V)
        out = np.empty((V, self.T), self.dtype)
        self.lib.qbo_signal_fwd_ex(C.byref(self.phys), self._p(y), self._p(h) if h is not None else None,
                                   self._p(a) if a is not None else None,
                                   f.ctypes.data_as(C.c_void_p) if f is not None else None,
                                   self._p(out), C.c_int64(V))
        return out

    def encoder_fwd(self, w, x):
        x = self._a(x)
        T = x.shape[-1]
        N = x.size // T
        ws, keep = self._weights(w)
        assert T == ws.T
        o1 = np.empty((N, 5), self.dtype)
        o2 = np.empty((N, 5), self.dtype)
        sg = np.empty((N, T), self.dtype)
        self.lib.qbo_encoder_fwd(C.byref(ws), C.byref(self.cfg), self._p(x), self._p(o1),
                                 self._p(o2), self._p(sg), C.c_int64(N))
        del keep
        return o1, o2, sg

    def encoder_fwd_spatial(self, w, x, ln=None, dropout_rate=0.0, dropout_seed=0):
        """x [B, X, Y, Z, T] -> (out2 [B,X,Y,Z,5], sigma [B,X,Y,Z,T]); needs 9-tap weights.
        ln [L, 4, U] (gamma1, beta1, gamma2, beta2 per block): use_layer_norm; dropout_rate / dropout_seed: a
        training-mode forward under the library's dropout stream (model.py:131-140)."""
        x = self._a(x)
        B, X, Y, Z, T = x.shape
        ws, keep = self._weights(w)
        assert ws.taps == 9 and T == ws.T
        o2 = np.empty((B, X, Y, Z, 5), self.dtype)
        sg = np.empty((B, X, Y, Z, T), self.dtype)
        lnp = None if ln is None else self._a(ln)
        self.lib.qbo_set_normalizer(self._p(lnp) if lnp is not None else None, C.c_double(dropout_rate),
                                    C.c_uint64(dropout_seed))
        try:
            self.lib.qbo_encoder_fwd_spatial(C.byref(ws), C.byref(self.cfg), self._p(x), B, X, Y, Z,
                                             self._p(o2), self._p(sg))
        finally:
            self.lib.qbo_set_normalizer(None, C.c_double(0.0), C.c_uint64(0))
        del keep
        return o2, sg

    def smoothness_loss(self, q, mask):
        """model.py:726-754 on q [B,X,Y,Z,5], mask [B,X,Y,Z]."""
        q = self._a(q)
        B, X, Y, Z, _ = q.shape
        mask = self._a(mask, (B, X, Y, Z))
        self.lib.qbo_smoothness_sum.restype = C.c_double
        s = self.lib.qbo_smoothness_sum(self._p(q), self._p(mask), B, X, Y, Z)
        return s / float(mask.sum())

    # -- logit-normal --------------------------------------------------------------------
    def reparam(self, q, z):
        q = self._a(q, (-1, 5))
        z = self._a(z, (-1, 2))
        out = np.empty((q.shape[0], 2), self.dtype)
        self.lib.qbo_reparam(self._p(q), self._p(z), self._p(out), C.c_int64(q.shape[0]))
        return out

    def logit_mvn_nlogp(self, y, p):
        y = self._a(y, (-1, 2))
        p = self._a(p, (-1, 5))
        out = np.empty(y.shape[0], self.dtype)
        self.lib.qbo_logit_mvn_nlogp(self._p(y), self._p(p), self._p(out), C.c_int64(y.shape[0]))
        return out

    def kl_diag(self, q, prior):
        q = self._a(q, (-1, 5))
        prior = self._a(prior, (-1, 5))
        out = np.empty(q.shape[0], self.dtype)
        self.lib.qbo_kl_diag(self._p(q), self._p(prior), self._p(out), C.c_int64(q.shape[0]))
        return out

    def population_prior_cost(self, prior4, batch):
        """model.py:710-716: the inverse-gamma(1, 2) cost on the population prior's log-variances x batch size."""
        p = self._a(prior4, (4,))
        self.lib.qbo_population_prior_cost.restype = C.c_double
        return self.lib.qbo_population_prior_cost(self._p(p), C.c_int(int(batch)))

    def kl_mog(self, q, comps, z):
        """model.py:666-685: one-draw KL estimate against a mixture-of-Gaussians population prior, comps [M, 4]."""
        q = self._a(q, (-1, 5))
        comps = self._a(comps, (-1, 4))
        z = self._a(z, (q.shape[0], 2))
        out = np.empty(q.shape[0], self.dtype)
        self.lib.qbo_kl_mog(self._p(q), self._p(comps), C.c_int(comps.shape[0]), self._p(z), self._p(out),
                            C.c_int64(q.shape[0]))
        return out

    def logit_gaussian_nlogp(self, y, p):
        y = self._a(y, (-1, 2))
        p = self._a(p, (-1, 5))
        out = np.empty(y.shape[0], self.dtype)
        self.lib.qbo_logit_gaussian_nlogp(self._p(y), self._p(p), self._p(out), C.c_int64(y.shape[0]))
        return out

    def synthetic_data_loss(self, y_true, q, inv_gamma_alpha=0.0, inv_gamma_beta=0.0):
        y = self._a(y_true, (-1, 3))
        q = self._a(q, (-1, 5))
        return self.lib.qbo_synthetic_data_loss_ig(self._p(y), self._p(q), C.c_double(inv_gamma_alpha),
                                                   C.c_double(inv_gamma_beta), C.c_int64(y.shape[0]))

    def nll(self, x, mask, pred, sigma):
        x = self._a(x)
        T = x.shape[-1]
        N = x.size // T
        mask = self._a(mask, (N,))
        pred = self._a(pred, (N, T))
        sigma = self._a(sigma, (N, T))
        out = np.empty(N, self.dtype)
        self.lib.qbo_nll(C.byref(self.cfg), self._p(x), self._p(mask), self._p(pred),
                         self._p(sigma), self._p(out), T, C.c_int64(N))
        return out

    def kl_samples(self, q, prior, z):
        q = self._a(q, (-1, 5))
        prior = self._a(prior, (-1, 5))
        N = q.shape[0]
        z = self._a(z, (N, -1, 2))
        out = np.empty(N, self.dtype)
        self.lib.qbo_kl_samples(self._p(q), self._p(prior), self._p(z), z.shape[1], self._p(out),
                                C.c_int64(N))
        return out

    def kl_closed(self, q, prior):
        q = self._a(q, (-1, 5))
        prior = self._a(prior, (-1, 5))
        out = np.empty(q.shape[0], self.dtype)
        self.lib.qbo_kl_closed(self._p(q), self._p(prior), self._p(out), C.c_int64(q.shape[0]))
        return out

    def moments(self, q, z):
        q = self._a(q, (-1, 5))
        N = q.shape[0]
        z = self._a(z, (N, -1, 2))
        means = np.empty((N, 3), self.dtype)
        var = np.empty((N, 3), self.dtype)
        self.lib.qbo_moments(C.byref(self.phys), self._p(q), self._p(z), z.shape[1],
                             self._p(means), self._p(var), C.c_int64(N))
        return means, var

    def elbo(self, x, mask, q, prior, sigma, zs, zk):
        """Returns dict(nll_v, kl_v, sums=(sum m*nll, sum kl[m>0], sum m), nll, kl, elbo)."""
        x = self._a(x)
        T = x.shape[-1]
        N = x.size // T
        assert T == self.T
        x = x.reshape(N, T)
        mask = self._a(mask, (N,))
        q = self._a(q, (N, 5))
        prior = self._a(prior, (N, 5))
        sigma = self._a(sigma, (N, T))
        zs = self._a(zs, (N, -1, 2))
        zk = self._a(zk, (N, -1, 2))
        nll_v = np.empty(N, self.dtype)
        kl_v = np.empty(N, self.dtype)
        sums = np.zeros(3, np.float64)
        self.lib.qbo_elbo(C.byref(self.phys), C.byref(self.cfg), self._p(x), self._p(mask),
                          self._p(q), self._p(prior), self._p(sigma), self._p(zs), zs.shape[1],
                          self._p(zk), zk.shape[1], self._p(nll_v), self._p(kl_v),
                          sums.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(N))
        nll = sums[0] / sums[2]
        kl = sums[1] / sums[2]
        return dict(nll_v=nll_v, kl_v=kl_v, sums=sums, nll=nll, kl=kl, elbo=nll + kl)

    # -- RNG -----------------------------------------------------------------------------
    def philox(self, ctr, key, rounds=10):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        {10: self.lib.qbo_philox4x32_10, 7: self.lib.qbo_philox4x32_7}[rounds](c, k, o)
        return tuple(int(v) for v in o)

    def philox_normals(self, seed, stream, voxel0, N, n):
        z = np.empty((N, n, 2), self.dtype)
        self.lib.qbo_philox_normals(C.c_uint64(seed), C.c_uint32(stream), C.c_int64(voxel0),
                                    C.c_int64(N), C.c_int(n), self._p(z))
        return z


def synth_inputs(N, params=None, seed=1, noise=True, oracle=None):
    """SURVEY 8(d) synthetic voxels: OEF = clip(N(0.4,0.2),0.05,0.8), DBV = TruncNormal(0.025,
    0.02; [0.003,0.195]) i.i.d. (config:48-58, uniform_prop=0), signals = forward model with the
    reference noise model for T=11 (signals.py:116-128)."""
    from scipy.stats import truncnorm
    rng = np.random.default_rng(seed)
    o = oracle or Oracle("f32", params)
    oef = np.clip(rng.standard_normal(N) * 0.2 + 0.4, 0.05, 0.8)
    a, b = (0.003 - 0.025) / 0.02, (0.195 - 0.025) / 0.02
    dbv = truncnorm.rvs(a, b, loc=0.025, scale=0.02, size=N, random_state=rng)
    y = np.stack([oef, dbv], -1).astype(np.float32)
    sig = o.signal_fwd(y).astype(np.float32)
    if noise and sig.shape[-1] == 11:
        norm_snr = np.array([0.985, 1.00, 1.01, 1., 0.97, 0.95, 0.93, 0.90, 0.86, 0.83, 0.79],
                            np.float32)
        snr = rng.uniform(50, 120, (N, 1)).astype(np.float32) * norm_snr[None]
        std = sig.mean(0, keepdims=True) / snr
        sig = sig + rng.standard_normal(sig.shape).astype(np.float32) * std
    return sig.astype(np.float32), y


def fit_wls(signals, params=None, tau_min=0.016):
    """loglinear.fit_wls (loglinear.py:68-105) restated in numpy float64: per voxel the weighted
    least-squares line through (tau, ln S) for taus > 0.016 with weights 1/tau -- what
    sklearn.LinearRegression.fit(X, Y, sample_weight=w) solves -- then R2' = -slope,
    DBV = intercept - ln S(tau=0), OEF = R2'/(DBV gamma 4/3 pi dchi hct B0), clipped.
    taus: np.around(np.arange(start, end, step, dtype=float32), 7) as :126-127.
    Returns (oef, dbv, r2p), each [..., 1]."""
    P = dict(DEFAULT_PARAMS if params is None else params)
    taus = np.around(np.arange(float(P["tau_start"]), float(P["tau_end"]), float(P["tau_step"]),
                               dtype=np.float32), decimals=7)
    signals = np.asarray(signals)
    with np.errstate(all="ignore"):
        ln_s = np.log(signals.astype(np.float64))
    ln_s[np.isnan(ln_s)] = 0
    ln_s[np.isinf(ln_s)] = 0
    line = np.where(taus > np.float32(tau_min))[0]
    x = taus[line].astype(np.float64)
    w = (1 / taus[line]).astype(np.float64)
    y = ln_s[..., line]
    xm = (w * x).sum() / w.sum()
    ym = (w * y).sum(-1, keepdims=True) / w.sum()
    slope = (w * (x - xm) * (y - ym)).sum(-1, keepdims=True) / (w * (x - xm) ** 2).sum()
    icpt = ym - slope * xm
    s0 = np.where(taus == 0)[0]
    r2p = -slope
    dbv = icpt - ln_s[..., s0]
    with np.errstate(all="ignore"):
        oef = r2p / (dbv * float(P["gamma"]) * (4 / 3) * np.pi * float(P["dchi"]) * float(P["hct"])
                     * float(P["b0"]))
    return np.clip(oef, 0.01, 0.8), np.clip(dbv, 0.002, 0.25), np.clip(r2p, 1e-2, 100)

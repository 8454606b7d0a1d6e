"""CPU tests pinning the oracle (oracle/qbold_oracle.c).

The reference has no tests or vectors for this path ("parity unpinned"), so the oracle
is pinned by (a) published known-answer vectors of the algorithms TensorFlow's kernels implement
(Random123 Philox KATs), (b) independent float64 mathematics (scipy Bessel / quadrature /
distributions, closed-form Gaussian KL, finite differences), (c) the surveyor's independent scipy
probes recorded in SURVEY.md 8c, and (d) the committed restatement goldens (drift guard).
"""
import os

import numpy as np
import pytest
import scipy.integrate as si
import scipy.special as sp
import scipy.stats as st

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "restatement_goldens.npz")


def test_philox_known_answer_vectors(oracle32):
    # Random123 kat_vectors, philox4x32-10 (the noise model's uniform draw keeps ten rounds)
    assert oracle32.philox((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert oracle32.philox((0xffffffff,) * 4, (0xffffffff,) * 2) == \
        (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert oracle32.philox((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
                           (0xa4093822, 0x299f31d0)) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_philox_seven_round_known_answer_vectors(oracle32):
    # Random123 kat_vectors, philox4x32 with 7 rounds: the rounds the normal stream runs since round 4
    assert oracle32.philox((0, 0, 0, 0), (0, 0), rounds=7) == (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)
    assert oracle32.philox((0xffffffff,) * 4, (0xffffffff,) * 2, rounds=7) == \
        (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)
    assert oracle32.philox((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
                           (0xa4093822, 0x299f31d0), rounds=7) == (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)


def test_philox_normals_are_standard_normal(oracle32):
    z = oracle32.philox_normals(7, 0, 0, 50000, 8).ravel()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert st.kstest(z[::7], "norm").pvalue > 1e-3
    # keyed by global voxel: a shifted window reproduces the same numbers
    a = oracle32.philox_normals(7, 1, 100, 10, 5)
    b = oracle32.philox_normals(7, 1, 0, 110, 5)[100:]
    assert np.array_equal(a, b)
    assert not np.array_equal(a, oracle32.philox_normals(8, 1, 100, 10, 5))
    # the first draws of a longer request are the draws of a shorter one (draw i = word i & 3 of call i >> 2)
    assert np.array_equal(oracle32.philox_normals(7, 1, 0, 10, 70)[:, :5], oracle32.philox_normals(7, 1, 0, 10, 5))


def test_philox_normal_stream_definition(oracle32):
    """The stream as qbold_dev.h defines it, recomputed here from the Philox words: the low sixteen bits give the radius,
    the top 23 the angle (its sixteen leading bits independent of the radius), four draws per Philox4x32-7 call keyed
    (voxel, call index, stream id; seed)."""
    seed, stream, vox = 0x123456789, 1, (1 << 33) + 17
    z = oracle32.philox_normals(seed, stream, vox, 1, 11)[0].astype(np.float64)
    for i in range(11):
        w = oracle32.philox((vox & 0xffffffff, vox >> 32, i >> 2, stream), (seed & 0xffffffff, seed >> 32), rounds=7)[i & 3]
        r = np.sqrt(-2.0 * np.log(((w & 0xffff) + 0.5) / 65536.0))
        th = 2.0 * np.pi * (w >> 9) / 2.0 ** 23
        assert abs(z[i, 0] - r * np.cos(th)) < 1e-6 and abs(z[i, 1] - r * np.sin(th)) < 1e-6


def test_philox_normals_moments_and_tails(oracle32):
    """Sixteen-bit radius and angle put the normals on a polar lattice; its moments, tails and independence against
    the Gaussian's on 3.2 M draws (tolerances: four standard errors of each estimator)."""
    z = oracle32.philox_normals(11, 0, 0, 400000, 8).astype(np.float64)
    n = z.size
    f = z.ravel()
    assert abs(f.mean()) < 4 / np.sqrt(n)
    assert abs(f.var() - 1.0) < 4 * np.sqrt(2.0 / n)
    assert abs((f ** 3).mean()) < 4 * np.sqrt(15.0 / n)
    assert abs((f ** 4).mean() - 3.0) < 4 * np.sqrt(96.0 / n)
    assert abs((f ** 6).mean() - 15.0) < 4 * np.sqrt((10395.0 - 225.0) / n)
    for t in (1.0, 2.0, 3.0, 4.0):
        p = 2.0 * st.norm.sf(t)
        assert abs(np.mean(np.abs(f) > t) - p) < 4 * np.sqrt(p / n), t
    assert np.abs(f).max() <= 4.8549 and np.abs(f).max() > 4.5      # |z| <= sqrt(-2 ln 2^-17)
    # the two normals of a draw, and consecutive draws of a voxel, are uncorrelated (also in their squares)
    a, b = z[..., 0].ravel(), z[..., 1].ravel()
    m = a.size
    assert abs(np.mean(a * b)) < 4 / np.sqrt(m) and abs(np.mean((a * a - 1) * (b * b - 1))) < 4 * 2 / np.sqrt(m)
    c, d = z[:, :-1, 0].ravel(), z[:, 1:, 0].ravel()
    assert abs(np.mean(c * d)) < 4 / np.sqrt(c.size) and abs(np.mean((c * c - 1) * (d * d - 1))) < 4 * 2 / np.sqrt(c.size)
    # radius^2 / 2 is exponential, the angle uniform
    r2 = 0.5 * (a * a + b * b)
    assert st.kstest(r2[::5], "expon").pvalue > 1e-3
    ang = np.arctan2(b, a) / (2 * np.pi) % 1.0
    assert st.kstest(ang[::5], "uniform").pvalue > 1e-3


def test_bessel_j0_is_cephes_single(oracle32, oracle64):
    x = np.linspace(0, 40, 400001)
    assert np.abs(oracle32.j0(x) - sp.j0(x)).max() < 1e-6      # Cephes j0f accuracy (f32 argument reduction at x ~ 40)
    assert np.abs(oracle64.j0(x) - sp.j0(x)).max() < 1e-14
    # small-argument branch 1 - z/4 (what makes Simpson node 0 vanish in float32)
    assert oracle32.j0(np.array([2.4e-4]))[0] == np.float32(1.0)


def test_tissue_integral(oracle32, oracle64):
    xs = np.array([0, .05, .1, .5, 1, 2, 4, 8, 16], np.float64)
    # surveyor's independent float64 Simpson probe
    want64 = [0, 7.49845523e-4, 2.99857724e-3, 7.43270461e-2, 2.89584437e-1, 1.04827512, 3.05363757,
              7.02366447, 15.0097522]
    np.testing.assert_allclose(oracle64.tissue_F(xs), want64, rtol=2e-8, atol=1e-12)
    # Simpson-129 vs adaptive quadrature of the same integrand: 129 nodes resolve it to ~1e-4
    def quad(x):
        return si.quad(lambda u: (2 + u) * np.sqrt(1 - u) * (1 - sp.j0(1.5 * x * u)) / (3 * u * u),
                       0, 1, limit=200)[0]
    np.testing.assert_allclose(oracle64.tissue_F(xs[1:]), [quad(x) for x in xs[1:]], rtol=2e-4)
    # float32 semantics: node 0 contributes 0 -> F32 = F64 - 9.77e-4 x^2
    d = oracle64.tissue_F(xs) - oracle32.tissue_F(xs).astype(np.float64)
    np.testing.assert_allclose(d, 9.7656e-4 * xs ** 2, atol=3e-5)
    # asymptotes of the log-linear model (signals.py:204-205): 0.3 x^2 and x - 1
    assert abs(oracle64.tissue_F(np.array([0.01]))[0] / 1e-4 - 0.3) < 1e-3
    assert abs(oracle64.tissue_F(np.array([40.0]))[0] - 39.0) < 0.35
    # derivative (J1 kernel) vs central differences
    x = np.array([0.3, 1.0, 3.0, 7.0, 12.0])
    # (node 0 carries a weight of 6.7e9: 1 - J0 there has ~1e-9 of cancellation noise even in
    # float64, so the step must not be small)
    fd = (oracle64.tissue_F(x + 2e-3) - oracle64.tissue_F(x - 2e-3)) / 4e-3
    np.testing.assert_allclose(oracle64.tissue_dF(x), fd, rtol=2e-5)
    np.testing.assert_allclose(oracle32.tissue_dF(x), fd, rtol=3e-5)


def test_signal_model_constants_and_probe(oracle32, oracle64):
    assert oracle32.T == 11 and oracle32.se_idx == 2
    np.testing.assert_allclose(oracle32.taus, -0.016 + 0.008 * np.arange(11), atol=1e-8)
    assert oracle32.taus[2] == 0.0
    # surveyor's float64 probe for (OEF, DBV) = (0.4, 0.12)
    want = [0.3757017, 0.4090463, 0.4224256, 0.4090463, 0.3757017, 0.3358403, 0.2987211, 0.2665907,
            0.2382160, 0.2124614, 0.1891701]
    np.testing.assert_allclose(oracle64.signal_fwd([[0.4, 0.12]])[0], want, rtol=5e-7)
    # independent recomputation of the whole model in numpy float64
    oef, dbv, te, r2t, hct = 0.4, 0.12, 0.074, 11.5, 0.34
    dw = (4 / 3) * np.pi * 2.67513e8 * 3.0 * 2.64e-7 * hct * oef
    assert abs(dw / oef - 301.74327499379774) < 1e-9
    taus = -0.016 + 0.008 * np.arange(11)
    F = oracle64.tissue_F(taus * dw)
    tissue = np.exp(-dbv * F) * np.exp(-te * r2t)
    m_bld = 1 - (2 - np.exp(-(3.0 - 1.21) / 1.58)) * np.exp(-1.21 / 1.58)
    assert abs(m_bld - 0.2198556289652348) < 1e-12
    td, r2b = (2.6 ** 2 / 2) * 1e-3, 1 / 0.189
    g0 = (4 / 45) * hct * (1 - hct) * (4 * np.pi * 3.0 * 2.64e-7 * oef) ** 2
    blood = np.exp(-r2b * te) * np.exp(-0.5 * 2.67513e8 ** 2 * g0 * td ** 2 * (
        te / td + np.sqrt(0.25 + te / td) + 1.5 - 2 * np.sqrt(0.25 + (te + taus) / td)
        - 2 * np.sqrt(0.25 + (te - taus) / td)))
    bw = m_bld * 0.775 * dbv
    np.testing.assert_allclose(oracle64.signal_fwd([[oef, dbv]])[0], (1 - bw) * tissue + bw * blood,
                               rtol=1e-7)  # taus are float32 in the oracle
    # float32 build agrees with float64 up to the node-0 artefact (<= ~5 % at the extremes)
    y = np.array([[0.4, 0.025], [0.04, 0.001], [0.84, 0.201]], np.float32)
    assert np.abs(oracle32.signal_fwd(y) / oracle64.signal_fwd(y) - 1).max() < 0.06


def test_signal_jacobian_finite_differences(oracle64):
    rng = np.random.default_rng(0)
    y = np.stack([rng.uniform(0.1, 0.8, 50), rng.uniform(0.01, 0.19, 50)], -1)
    jac = oracle64.signal_jac(y)
    for k in range(2):
        d = np.zeros_like(y)
        d[:, k] = 1e-4
        fd = (oracle64.signal_fwd(y + d) - oracle64.signal_fwd(y - d)) / 2e-4
        np.testing.assert_allclose(jac[:, :, k], fd, rtol=2e-4, atol=2e-6)
    from oracle.oracle import Oracle
    for full, blood in ((False, True), (True, False)):
        o = Oracle("f64", full_model=full, include_blood=blood)
        jac = o.signal_jac(y)
        d = np.zeros_like(y)
        d[:, 1] = 1e-4
        fd = (o.signal_fwd(y + d) - o.signal_fwd(y - d)) / 2e-4
        np.testing.assert_allclose(jac[:, :, 1], fd, rtol=2e-4, atol=2e-6)


def test_logit_normal_density(oracle64, oracle32):
    rng = np.random.default_rng(1)
    n = 200
    p = rng.normal(size=(n, 5))
    y = np.stack([rng.uniform(0.05, 0.83, n), rng.uniform(0.002, 0.2, n)], -1)
    so, sd = np.tanh(p[:, 1]) * 3 - 1, np.tanh(p[:, 3]) * 3 - 1
    c = np.tanh(p[:, 4]) * np.exp(-2.0)
    x = np.stack([(y[:, 0] - 0.04) / 0.8, (y[:, 1] - 0.001) / 0.2], -1)
    lg = np.log(x / (1 - x))
    want = np.empty(n)
    for i in range(n):
        L = np.array([[np.exp(so[i]), 0], [c[i], np.exp(sd[i])]])
        mvn = st.multivariate_normal([p[i, 0], p[i, 2]], L @ L.T)
        # the reference adds +sum log x(1-x) to the negative Gaussian log-density (model.py:398)
        want[i] = -mvn.logpdf(lg[i]) + np.sum(np.log(x[i]) + np.log(1 - x[i]))
    np.testing.assert_allclose(oracle64.logit_mvn_nlogp(y, p), want, rtol=1e-9, atol=1e-9)
    got32 = oracle32.logit_mvn_nlogp(y, p)
    assert np.max(np.abs(got32 - want) / (np.abs(want) + 1)) < 1e-4
    # reparameterised samples follow the stated Cholesky factor
    q = np.tile(np.array([[0.3, 0.2, -1.0, -0.1, 0.8]]), (100000, 1))
    z = rng.standard_normal((100000, 2))
    s = oracle64.reparam(q, z)
    lgs = np.log(np.stack([(s[:, 0] - 0.04) / 0.8, (s[:, 1] - 0.001) / 0.2], -1))
    lgs = lgs - np.log(1 - np.exp(lgs))
    so0, sd0, c0 = np.tanh(0.2) * 3 - 1, np.tanh(-0.1) * 3 - 1, np.tanh(0.8) * np.exp(-2)
    cov = np.cov(lgs.T)
    np.testing.assert_allclose(cov, [[np.exp(2 * so0), c0 * np.exp(so0)],
                                     [c0 * np.exp(so0), np.exp(2 * sd0) + c0 ** 2]], rtol=0.03, atol=2e-3)


def test_kl_estimators(oracle64):
    rng = np.random.default_rng(2)
    n = 64
    q, p = rng.normal(size=(n, 5)) * 0.5, rng.normal(size=(n, 5)) * 0.5
    q[:, 4] = p[:, 4] = -30.0   # tanh -> -1: fixed covariance term; then the same for both
    z = rng.standard_normal((n, 20000, 2))
    mc = oracle64.kl_samples(q, p, z)

    def gauss_kl(q, p):
        out = np.empty(len(q))
        for i in range(len(q)):
            def ch(v):
                L = np.array([[np.exp(np.tanh(v[1]) * 3 - 1), 0],
                              [np.tanh(v[4]) * np.exp(-2), np.exp(np.tanh(v[3]) * 3 - 1)]])
                return np.array([v[0], v[2]]), L @ L.T
            mq, Sq = ch(q[i])
            mp, Sp = ch(p[i])
            Spi = np.linalg.inv(Sp)
            d = mp - mq
            out[i] = 0.5 * (np.trace(Spi @ Sq) + d @ Spi @ d - 2 + np.log(np.linalg.det(Sp) / np.linalg.det(Sq)))
        return out
    np.testing.assert_allclose(mc, gauss_kl(q, p), rtol=0.08, atol=0.02)
    # reference closed form (model.py:612-652) is exact when the covariance term vanishes
    q0, p0 = q.copy(), p.copy()
    q0[:, 4] = p0[:, 4] = 0.0
    np.testing.assert_allclose(oracle64.kl_closed(q0, p0), gauss_kl(q0, p0), rtol=1e-9, atol=1e-10)
    assert np.all(oracle64.kl_closed(q0, q0) < 1e-12)


def test_nll_branches(params):
    from oracle.oracle import Oracle
    rng = np.random.default_rng(3)
    n, T = 50, 11
    x = rng.uniform(0.2, 1.0, (n, T))
    pred = rng.uniform(0.2, 1.0, (n, T))
    sigma = rng.uniform(0.02, 0.2, (n, T))
    mask = np.ones(n)
    g = Oracle("f64", params)
    yt, yp = x / (x[:, 2:3] + 1e-3), pred / (pred[:, 2:3] + 1e-3)
    want = -st.norm.logpdf(yt - yp, scale=sigma).sum(-1)
    np.testing.assert_allclose(g.nll(x, mask, pred, sigma), want, rtol=1e-10)
    t = Oracle("f64", params, student_t_df=3.0)
    np.testing.assert_allclose(t.nll(x, mask, pred, sigma),
                               -st.t.logpdf(yt - yp, 3.0, scale=sigma).sum(-1), rtol=1e-10)
    big_df = Oracle("f64", params, student_t_df=200)  # >= 50 -> Gaussian (Appendix B1)
    np.testing.assert_allclose(big_df.nll(x, mask, pred, sigma), want, rtol=1e-10)
    m = Oracle("f64", params, multi_image_normalisation=True, predict_log_data=True)
    yt = np.log(x / (x[:, 1:4].mean(-1, keepdims=True) + 1e-3))
    yp = np.log(pred / (pred[:, 1:4].mean(-1, keepdims=True) + 1e-3))
    np.testing.assert_allclose(m.nll(x, mask, pred, sigma),
                               -st.norm.logpdf(yt - yp, scale=sigma).sum(-1), rtol=1e-10)


def test_encoder_matches_numpy(params):
    from oracle.oracle import Oracle, init_weights
    o = Oracle("f64", params)
    for cw in (True, False):
        w = init_weights(T=11, U=12, L=2, channelwise_gating=cw, seed=5)
        rng = np.random.default_rng(5)
        for k in ("b0", "bc", "br1", "br2", "bg", "bf"):
            w[k] = rng.normal(size=w[k].shape).astype(np.float32) * 0.1
        w["gate_offset"] = -1.5
        x = rng.uniform(0.05, 1.0, (20, 11))
        n = np.log(np.clip(x, 1e-2, 1e8) / np.clip(x, 1e-2, 1e8)[:, 2:3])
        W = {k: np.asarray(v, np.float64) for k, v in w.items() if k not in ("meta", "gate_offset")}
        relu = lambda v: np.maximum(v, 0)
        a = b = relu(n @ W["W0"] + W["b0"])
        for l in range(2):
            a = relu(a @ W["Wc"][l] + W["bc"][l])
            skip = relu(b @ W["Wc"][l] + W["bc"][l])
            r = relu(relu(b) @ W["Wr1"][l] + W["br1"][l]) @ W["Wr2"][l] + W["br2"][l]
            g = 1 / (1 + np.exp(-(r @ W["Wg"][l] + W["bg"][l] - 1.5)))
            b = skip * (1 - g) + r * g
        o1, o2, sg = o.encoder_fwd(w, x)
        np.testing.assert_allclose(o1, a @ W["Wf"] + W["bf"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o2, b @ W["Wf"] + W["bf"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(sg, np.exp(b @ W["Ws"] + W["bs"]), rtol=1e-9)


def test_spatial_encoder_matches_scipy_correlate(params):
    """Keras Conv3D((3,3,1), padding='same') is a cross-correlation with zero padding: out[x,y] =
    sum_{i,j} in[x+i-1, y+j-1] @ K[i,j] (model.py:152-157).  Checked against scipy.ndimage.correlate
    channel by channel; the TV term (model.py:726-754) against a numpy restatement."""
    import scipy.ndimage as ndi
    from oracle.oracle import Oracle, init_weights
    o = Oracle("f64", params)
    w = init_weights(T=11, U=8, L=2, channelwise_gating=True, seed=7, taps=9)
    rng = np.random.default_rng(7)
    for k in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[k] = rng.normal(size=w[k].shape).astype(np.float32) * 0.1
    w["Wr1"] = (w["Wr1"] * 6).astype(np.float32)
    w["Wr2"] = (w["Wr2"] * 6).astype(np.float32)
    w["gate_offset"] = 0.3
    B, X, Y, Z = 2, 5, 4, 3
    x = rng.uniform(0.05, 1.0, (B, X, Y, Z, 11))
    W = {k: np.asarray(v, np.float64) for k, v in w.items() if k not in ("meta", "gate_offset")}
    relu = lambda v: np.maximum(v, 0)

    def conv(a, K, bias):  # a [B,X,Y,Z,Cin], K [3,3,Cin,Cout]
        out = np.zeros(a.shape[:-1] + (K.shape[-1],))
        for co in range(K.shape[-1]):
            for ci in range(K.shape[-2]):
                k5 = K[:, :, ci, co][None, :, :, None]
                out[..., co] += ndi.correlate(a[..., ci], k5, mode="constant", cval=0.0)
        return out + bias

    xc = np.clip(x, 1e-2, 1e8)
    b = relu(np.log(xc / xc[..., 2:3]) @ W["W0"] + W["b0"])
    for l in range(2):
        skip = relu(b @ W["Wc"][l] + W["bc"][l])
        r = conv(relu(conv(relu(b), W["Wr1"][l], W["br1"][l])), W["Wr2"][l], W["br2"][l])
        g = 1 / (1 + np.exp(-(r @ W["Wg"][l] + W["bg"][l] + 0.3)))
        b = skip * (1 - g) + r * g
    o2, sg = o.encoder_fwd_spatial(w, x)
    np.testing.assert_allclose(o2, b @ W["Wf"] + W["bf"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(sg, np.exp(b @ W["Ws"] + W["bs"]), rtol=1e-9)
    # a 1x1 crop sees only the centre tap: the voxel-wise encoder with centre-tap weights
    wc = dict(w, Wr1=w["Wr1"][:, 1, 1], Wr2=w["Wr2"][:, 1, 1], meta=dict(w["meta"], taps=1))
    x1 = x.reshape(-1, 1, 1, 1, 11)
    np.testing.assert_allclose(o.encoder_fwd_spatial(w, x1)[0].reshape(-1, 5), o.encoder_fwd(wc, x1)[1],
                               rtol=1e-10, atol=1e-13)

    q = rng.normal(size=(B, X, Y, Z, 5))
    mask = (rng.uniform(size=(B, X, Y, Z)) > 0.3).astype(np.float64)
    sig = lambda v: 1 / (1 + np.exp(-v))
    oef, dbv = sig(q[..., 0]) * 0.8 + 0.04, sig(q[..., 2]) * 0.2 + 0.001   # forward_transform
    p = np.stack([oef / 0.8, dbv / 0.2], -1)                                  # model.py:735-736
    dx = np.abs(p[:, 1:] - p[:, :-1]).sum(-1) * (mask[:, 1:] * mask[:, :-1] > 0)
    dy = np.abs(p[:, :, 1:] - p[:, :, :-1]).sum(-1) * (mask[:, :, 1:] * mask[:, :, :-1] > 0)
    np.testing.assert_allclose(o.smoothness_loss(q, mask), (dx.sum() + dy.sum()) / mask.sum(), rtol=1e-10)


def test_wls_restatement_matches_sklearn(oracle32, params):
    """loglinear.fit_wls calls sklearn.linear_model.LinearRegression per voxel (loglinear.py:80-91);
    sklearn is installed here, so the closed form in oracle.fit_wls is pinned to the reference's own
    dependency on the reference's own call pattern."""
    from sklearn.linear_model import LinearRegression
    from oracle.oracle import fit_wls
    rng = np.random.default_rng(3)
    y = np.stack([rng.uniform(0.1, 0.7, 40), rng.uniform(0.01, 0.12, 40)], -1)
    sig = oracle32.signal_fwd(y).astype(np.float64) * 300.0 * (1 + 0.01 * rng.normal(size=(40, 11)))
    sig[3, 7] = 0.0        # ln -> -inf -> 0
    sig[4, 9] = -1.0       # ln -> nan -> 0
    taus = np.around(np.arange(-0.016, 0.065, 0.008, dtype=np.float32), decimals=7)
    line = np.where(taus > 0.016)
    assert list(line[0]) == [5, 6, 7, 8, 9, 10]
    w = 1 / taus[line]
    with np.errstate(all="ignore"):
        ln_s = np.log(sig)
    ln_s[np.isnan(ln_s)] = 0
    ln_s[np.isinf(ln_s)] = 0
    p = np.zeros((40, 2))
    for v in range(40):
        X = np.vstack((taus[line], np.ones_like(taus[line]))).T
        wls = LinearRegression()
        wls.fit(X, np.squeeze(ln_s[v, line]), sample_weight=w)
        p[v] = [wls.coef_[0], wls.intercept_]
    r2p = -p[:, 0:1]
    dbv = p[:, 1:] - ln_s[:, np.where(taus == 0)[0]]
    k = float(params["gamma"]) * (4 / 3) * np.pi * float(params["dchi"]) * float(params["hct"]) * float(params["b0"])
    oef = r2p / (dbv * k)
    o, d, r = fit_wls(sig.reshape(5, 2, 2, 2, 11), params)
    np.testing.assert_allclose(r.reshape(-1, 1), np.clip(r2p, 1e-2, 100), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(d.reshape(-1, 1), np.clip(dbv, 0.002, 0.25), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o.reshape(-1, 1), np.clip(oef, 0.01, 0.8), rtol=2e-3, atol=1e-6)
    # and the estimator is meaningful: noiseless R2' = dw * dbv recovered to ~20 %
    o2, d2, r2 = fit_wls(oracle32.signal_fwd(y), params)
    dw = float(params["gamma"]) * 4 / 3 * np.pi * float(params["dchi"]) * float(params["hct"]) * float(params["b0"]) * y[:, 0]
    assert np.median(np.abs(r2[:, 0] / (dw * y[:, 1]) - 1)) < 0.25


def test_inverse_gamma_prior_term(oracle64):
    """model.py:492-507 with fixed alpha / beta against scipy.stats.invgamma (= tfp InverseGamma)."""
    rng = np.random.default_rng(11)
    n = 50
    q = rng.normal(size=(n, 5)) * 0.5
    y = np.stack([rng.uniform(0.1, 0.7, n), rng.uniform(0.01, 0.15, n), np.ones(n)], -1)
    base = oracle64.synthetic_data_loss(y, q)
    assert oracle64.synthetic_data_loss(y, q, 3.0, 0.0) == base       # alpha * beta > 0 gates the term
    a, b = 3.0, 0.15
    so, sd = np.tanh(q[:, 1]) * 3 - 1, np.tanh(q[:, 3]) * 3 - 1
    oef_var, dbv_var = np.exp(so) ** 2, np.exp(sd) ** 2 + q[:, 4] ** 2
    prior = st.invgamma.logpdf(oef_var, a, scale=b) + st.invgamma.logpdf(dbv_var, a, scale=b)
    np.testing.assert_allclose(oracle64.synthetic_data_loss(y, q, a, b), base - prior.mean(), rtol=1e-12)


def test_diagonal_family_restatements(oracle64):
    """use_mvg=False (the reference's argparse default): closed-form KL = sum of two univariate Gaussian
    KLs (what tfp LogitNormal.kl_divergence returns), and logit_gaussian_log_prob = the product of two
    scipy logit-normal densities without the (2 pi)^-1 the reference leaves out (model.py:403-421)."""
    rng = np.random.default_rng(13)
    n = 200
    q, p = rng.normal(size=(n, 5)) * 0.6, rng.normal(size=(n, 5)) * 0.6
    want = 0.0
    for d in (0, 2):
        sq, sp = np.exp(np.tanh(q[:, d + 1]) * 3 - 1), np.exp(np.tanh(p[:, d + 1]) * 3 - 1)
        want = want + np.log(sp / sq) + (sq ** 2 + (q[:, d] - p[:, d]) ** 2) / (2 * sp ** 2) - 0.5
    np.testing.assert_allclose(oracle64.kl_diag(q, p), want, rtol=1e-10, atol=1e-12)
    # the Cholesky column is ignored
    q2 = q.copy(); q2[:, 4] += 1.0
    np.testing.assert_array_equal(oracle64.kl_diag(q2, p), oracle64.kl_diag(q, p))
    # with a zero Cholesky term the 5-parameter closed form (model.py:612-652) is exact and agrees
    q0, p0 = q.copy(), p.copy(); q0[:, 4] = 0; p0[:, 4] = 0
    np.testing.assert_allclose(oracle64.kl_closed(q0, p0), want, rtol=1e-9, atol=1e-11)
    y = np.stack([rng.uniform(0.06, 0.8, n), rng.uniform(0.003, 0.19, n)], -1)
    x0, x1 = (y[:, 0] - 0.04) / 0.8, (y[:, 1] - 0.001) / 0.2
    lg = lambda x: np.log(x) - np.log1p(-x)
    so, sd = np.tanh(q[:, 1]) * 3 - 1, np.tanh(q[:, 3]) * 3 - 1
    # density of logit-normal x: N(logit x; mu, s) / (x (1 - x)); the reference ADDS log(x(1-x)) to
    # the negative log-density (:418) -- restated as written
    nl = -(st.norm.logpdf(lg(x0), q[:, 0], np.exp(so)) + st.norm.logpdf(lg(x1), q[:, 2], np.exp(sd))) \
        - np.log(2 * np.pi) + np.log(x0 * (1 - x0)) + np.log(x1 * (1 - x1))
    np.testing.assert_allclose(oracle64.logit_gaussian_nlogp(y, q), nl, rtol=1e-10)
    # = the 5-parameter density with a zero Cholesky term, less log 2 pi
    np.testing.assert_allclose(oracle64.logit_gaussian_nlogp(y, q), oracle64.logit_mvn_nlogp(y, q0) - np.log(2 * np.pi),
                               rtol=1e-10)


def test_moments_and_elbo_composition(oracle64):
    rng = np.random.default_rng(4)
    n, S, K = 30, 3, 5
    q, prior = rng.normal(size=(n, 5)) * 0.4, rng.normal(size=(n, 5)) * 0.4
    z = rng.standard_normal((n, 11, 2))
    means, var = oracle64.moments(q, z)
    s = np.stack([oracle64.reparam(q, z[:, k]) for k in range(11)], 1)
    np.testing.assert_allclose(means[:, :2], s.mean(1), rtol=1e-12)
    np.testing.assert_allclose(var[:, :2], s.var(1), rtol=1e-10)
    r2p = 301.74327499379774 * s[..., 0] * s[..., 1]
    np.testing.assert_allclose(means[:, 2], r2p.mean(1), rtol=1e-9)
    x = rng.uniform(0.2, 1.0, (n, 11))
    sigma = rng.uniform(0.02, 0.2, (n, 11))
    mask = (rng.uniform(size=n) > 0.3).astype(float)
    zs, zk = rng.standard_normal((n, S, 2)), rng.standard_normal((n, K, 2))
    e = oracle64.elbo(x, mask, q, prior, sigma, zs, zk)
    nll = np.zeros(n)
    for k in range(S):
        pred = oracle64.signal_fwd(oracle64.reparam(q, zs[:, k]))
        nll += oracle64.nll(x, mask, pred, sigma)
    np.testing.assert_allclose(e["nll_v"], nll / S, rtol=1e-12)
    np.testing.assert_allclose(e["kl_v"], oracle64.kl_samples(q, prior, zk), rtol=1e-12)
    assert abs(e["elbo"] - ((nll / S * mask).sum() + (e["kl_v"] * (mask > 0)).sum()) / mask.sum()) < 1e-10


def test_restatement_goldens_have_not_drifted(oracle32, oracle64):
    from oracle.oracle import WEIGHT_NAMES
    g = np.load(GOLD)
    np.testing.assert_array_equal(oracle32.tissue_F(g["F_x"]), g["F_f32"])
    np.testing.assert_allclose(oracle64.tissue_F(g["F_x"]), g["F_f64"], rtol=1e-13)
    np.testing.assert_allclose(oracle32.signal_fwd(g["sig_pts"]), g["sig_f32"], rtol=1e-6)
    w = {k: g["w_" + k] for k in WEIGHT_NAMES}
    w["meta"] = dict(T=11, U=60, L=2, channelwise_gating=True)
    w["gate_offset"] = -3.0
    prior, q, sigma = oracle32.encoder_fwd(w, g["x"])
    np.testing.assert_allclose(q, g["q"], rtol=1e-5, atol=1e-6)
    e = oracle32.elbo(g["x"], g["mask"], g["q"], g["prior"], g["sigma"], g["zs"], g["zk"])
    np.testing.assert_allclose(e["nll_v"], g["nll_v"], rtol=1e-5)
    assert abs(e["elbo"] - float(g["elbo"])) < 1e-6 * abs(float(g["elbo"]))
    np.testing.assert_allclose(oracle32.philox_normals(1, 0, 5, 8, 6), g["philox_z"], rtol=1e-6, atol=1e-7)
    # widening rows: crops, signal-model options, WLS
    from oracle.oracle import fit_wls
    w9 = {k: g["w9_" + k] for k in WEIGHT_NAMES}
    w9["meta"] = dict(T=11, U=12, L=2, channelwise_gating=True, taps=9)
    w9["gate_offset"] = -1.0
    sp_q, sp_sigma = oracle32.encoder_fwd_spatial(w9, g["crop"])
    np.testing.assert_allclose(sp_q, g["sp_q"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sp_sigma, g["sp_sigma"], rtol=1e-5)
    assert abs(oracle32.smoothness_loss(g["sp_q"], g["crop_mask"]) - float(g["tv"])) < 1e-6 * float(g["tv"])
    np.testing.assert_allclose(oracle32.signal_fwd_ex(g["ex_y"], hct=g["ex_hct"], alt=g["ex_alt"], from_idx=g["ex_idx"]),
                               g["ex_sig"], rtol=1e-6)
    wo, wd, wr = fit_wls(g["x"][:32].astype(np.float64) * 200.0)
    np.testing.assert_allclose(wo, g["wls_oef"], rtol=1e-9)
    np.testing.assert_allclose(wd, g["wls_dbv"], rtol=1e-9)
    np.testing.assert_allclose(wr, g["wls_r2p"], rtol=1e-9)


def test_two_independent_restatements_agree_on_config_1(oracle32, params):
    """BASELINE config 1 (optimal.yaml shapes, 4,096 synthetic voxels x 11 tau, CPU): oracle/torch_ref.py -- written
    separately from qbold_oracle.c, from the same reference lines, as whole-batch torch float32 ops at the
    reference's granularity ([V, T, 129] Bessel tensor, S-fold tiled batch) -- against the C restatement.  With no
    reference-held vector to pin either (TensorFlow cannot run here), two restatements agreeing to float32
    rounding is the tightest check available: signals 1e-6, encoder 2e-6, ELBO 1e-6 relative."""
    torch = pytest.importorskip("torch")
    from oracle import torch_ref as tr
    from oracle.oracle import init_weights, synth_inputs
    n, S, K = 4096, 2, 70
    w = init_weights(T=11, U=60, L=2, seed=1)
    w["gate_offset"] = -3.0
    rb = np.random.default_rng(0)
    for nm in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[nm] = (rb.standard_normal(w[nm].shape) * 0.1).astype(np.float32)
    x, y = synth_inputs(n, params, seed=1, oracle=oracle32)
    np.testing.assert_array_equal(tr.tau_grid(params).numpy(), oracle32.taus)
    sig_c, sig_t = oracle32.signal_fwd(y), tr.signal_model(y, params).numpy()
    assert np.max(np.abs(sig_c - sig_t) / np.abs(sig_c)) < 1e-6
    p1, q2, sg = oracle32.encoder_fwd(w, x)
    a, b, c = (t.numpy() for t in tr.encoder(w, x, oracle32.se_idx, -3.0))
    assert np.abs(a - p1).max() < 2e-6 and np.abs(b - q2).max() < 2e-6 and np.max(np.abs(c - sg) / sg) < 2e-6
    mask = (rb.uniform(size=n) > 0.1).astype(np.float32)
    zs, zk = oracle32.philox_normals(1, 0, 0, n, S), oracle32.philox_normals(1, 1, 0, n, K)
    want = oracle32.elbo(x, mask, q2, p1, sg, zs, zk)
    got = tr.elbo(x, mask, q2, p1, sg, zs, zk, params, oracle32.se_idx)
    assert abs(want["elbo"] - got["elbo"]) < 1e-6 * abs(want["elbo"])
    assert abs(want["nll"] - got["nll"]) < 1e-6 * abs(want["nll"]) and abs(want["kl"] - got["kl"]) < 1e-5 * abs(want["kl"])
    assert np.max(np.abs(want["nll_v"] - got["nll_v"]) / (np.abs(want["nll_v"]) + 1.0)) < 5e-5
    assert np.max(np.abs(want["kl_v"] - got["kl_v"]) / (np.abs(want["kl_v"]) + 1.0)) < 1e-5
    # the log-linear and no-blood signal models as well
    for full, blood in ((False, True), (True, False)):
        from oracle.oracle import Oracle
        oc = Oracle("f32", params, full_model=full, include_blood=blood)
        s_c, s_t = oc.signal_fwd(y[:512]), tr.signal_model(y[:512], params, full, blood).numpy()
        assert np.max(np.abs(s_c - s_t) / np.abs(s_c)) < 2e-6


def test_kl_draw_count_under_the_tiled_batch(oracle32, params):
    """With no_samples = S > 1 the reference concatenates S copies of the batch (model.py:245-246, 656) and draws K
    KL samples per COPY: S * K draws per voxel, averaged.  The torch restatement does exactly that (tiled rows);
    the kernels / C oracle never tile and draw S * K per voxel instead -- the same numbers when fed the same
    normals, which is what EncoderTrainer.kl_draws() asks of them by default."""
    torch = pytest.importorskip("torch")
    from oracle import torch_ref as tr
    from oracle.oracle import init_weights, synth_inputs
    n, S, K = 256, 3, 10
    w = init_weights(T=11, U=16, L=1, seed=5)
    w["gate_offset"] = -3.0
    x, _ = synth_inputs(n, params, seed=4, oracle=oracle32)
    p1, q2, sg = oracle32.encoder_fwd(w, x)
    mask = np.ones(n, np.float32)
    zs, zk = oracle32.philox_normals(2, 0, 0, n, S), oracle32.philox_normals(2, 1, 0, n, S * K)
    tiled = tr.elbo(x, mask, q2, p1, sg, zs, zk, params, oracle32.se_idx, kl_tiled=True)
    flat = oracle32.elbo(x, mask, q2, p1, sg, zs, zk)          # S * K draws per voxel, no tiling
    assert np.max(np.abs(tiled["kl_v"] - flat["kl_v"]) / (np.abs(flat["kl_v"]) + 1.0)) < 1e-5
    assert abs(tiled["kl"] - flat["kl"]) < 1e-5 * abs(flat["kl"])
    # K draws per voxel (kl_tiled=False) estimate the same KL with S times fewer draws
    few = oracle32.elbo(x, mask, q2, p1, sg, zs, zk[:, :K])
    assert abs(few["kl"] - flat["kl"]) < 6.0 * np.std(few["kl_v"] - flat["kl_v"]) / np.sqrt(n) + 1e-6

"""The CPU oracle (oracle/qbold_oracle.c) held to fixtures produced by the REFERENCE'S OWN SOURCE TEXT.

tests/golden/reference_text_goldens.npz was written by tests/golden/make_reference_text_goldens.py, which imports
/root/reference/{signals,model,logit_mvn}.py unmodified over a NumPy stand-in for TensorFlow (tests/golden/
tf_standin/).  These tests check WIRING -- layouts, channel orders, signs, masks, the S-fold tiled batch, which
layer feeds which -- not TensorFlow's arithmetic: the stand-in computes in NumPy float32, so tolerances here are a
few float32 ulps of the quantities involved (1e-5 relative, looser where a sum of O(100) terms is compared), and a
misread index or sign would miss them by orders of magnitude.  Parity with TensorFlow itself stays unpinned
(DESIGN section 2)."""
import os

import numpy as np
import pytest
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_text_goldens.npz"))


def g(case, *names):
    out = [G[f"{case}/{n}"] for n in names]
    return out[0] if len(out) == 1 else out


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))      # e.g. logit(0) = -inf on both sides
    a, b = np.where(same_inf, 0.0, a), np.where(same_inf, 0.0, b)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert np.all(err <= 0), f"max excess {err.max():.3e}; max |diff| {np.abs(a - b).max():.3e}"


def weights_of(case):
    from oracle.oracle import WEIGHT_NAMES
    w = {n: G[f"{case}/weights/{n}"] for n in WEIGHT_NAMES}
    U, L = int(G[f"{case}/weights/U"]), int(G[f"{case}/weights/L"])
    w["gate_offset"] = float(G[f"{case}/weights/gate_offset"])
    w["meta"] = dict(T=11, U=U, L=L, channelwise_gating=bool(G[f"{case}/weights/channelwise_gating"]), taps=9)
    return w


def centre_taps(w):
    c = dict(w)
    c["Wr1"], c["Wr2"] = w["Wr1"][:, 1, 1], w["Wr2"][:, 1, 1]
    c["meta"] = dict(w["meta"], taps=1)
    return c


# -- signals.py ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,full,blood", [("full_blood", True, True), ("full_noblood", True, False),
                                             ("loglinear_blood", False, True), ("loglinear_noblood", False, False)])
def test_forward_model_variants(params, case, full, blood):
    from oracle.oracle import Oracle
    o = Oracle("f32", params, full_model=full, include_blood=blood)
    y, s, taus = g(f"signal/{case}", "oef_dbv", "signal", "taus")
    np.testing.assert_array_equal(o.taus, taus)          # tf.range in float32, multiply form
    close(o.signal_fwd(y), s, rtol=2e-5)


def test_forward_model_shapes_protocol_hct(params, oracle32):
    from oracle.oracle import Oracle
    y5, s5 = g("signal/shape5d", "oef_dbv", "signal")
    assert s5.shape == y5.shape[:-1] + (11,)
    close(oracle32.signal_fwd(y5), s5, rtol=2e-5)
    y, dw, r2p = g("signal/dw_r2p", "oef_dbv", "dw", "r2p")
    close(dw / y[:, 0], np.full(len(y), 301.74327499379774), rtol=1e-6)
    close(r2p, dw * y[:, 1], rtol=1e-6)
    o24 = Oracle("f32", dict(params, tau_start="-0.028", tau_end="0.065", tau_step="0.004"))
    y, s, taus = g("signal/protocol24", "oef_dbv", "signal", "taus")
    assert o24.T == 24 and len(taus) == 24
    np.testing.assert_array_equal(o24.taus, taus)
    close(o24.signal_fwd(y), s, rtol=2e-5)
    yh, sh = g("signal/variable_hct", "oef_dbv_hct", "signal")
    close(oracle32.signal_fwd_ex(yh[:, :2], hct=yh[:, 2]), sh, rtol=2e-5)


def test_misalignment_augmentation(oracle32):
    """signals.py:80-96 with its draws made explicit: images AFTER from_index of a misaligned voxel come from the
    perturbed, clipped parameters."""
    y, s, u, idx, zo, zd, prob = g("signal/misaligned", "oef_dbv", "signal", "u_misaligned", "from_index", "z_oef",
                                    "z_dbv", "prob")
    mis = u[:, 0] < prob
    assert 20 < mis.sum() < 108 and idx.min() >= 4 and idx.max() <= 9      # uniform int in [4, T - 1)
    alt = np.stack([np.clip(zo[:, 0] * np.float32(0.15) + y[:, 0], 0.05, 0.8),
                    np.clip(zd[:, 0] * np.float32(0.05) + y[:, 1], 0.002, 0.3)], -1).astype(np.float32)
    from_idx = np.where(mis, idx[:, 0], 11).astype(np.int32)
    close(oracle32.signal_fwd_ex(y, alt=alt, from_idx=from_idx), s, rtol=2e-5)
    assert np.abs(oracle32.signal_fwd(y) - s).max() > 1e-2                    # the augmentation did something


def test_noise_model_11_taus(oracle32):
    """signals.py:116-128: std = batch mean per tau / (U(50, 120) * norm_snr)."""
    y, s, u, z = g("signal/noise11", "oef_dbv", "signal", "snr_uniform", "z")
    clean = oracle32.signal_fwd(y)
    norm_snr = np.array([0.985, 1.00, 1.01, 1., 0.97, 0.95, 0.93, 0.90, 0.86, 0.83, 0.79], np.float32)
    want = clean + z * (clean.mean(0, keepdims=True) / (u * norm_snr[None]))
    assert 50 <= u.min() and u.max() <= 120
    close(want, s, rtol=3e-5)


def test_create_synthetic_dataset_layout(oracle32, params):
    x, y, ou, on, du, dt, perm = g("synthetic_dataset", "x", "y", "oef_uniform", "oef_normal", "dbv_uniform",
                                    "dbv_truncnorm", "permutation")
    assert x.shape == (900, 11) and y.shape == (900, 3) and len(ou) == 3 and len(on) == 27
    oefs = np.concatenate([ou, np.clip(on * np.float32(params["oef_std"]) + np.float32(params["oef_mean"]),
                                       float(params["oef_start"]), float(params["oef_end"]))]).astype(np.float32)
    dbvs = np.concatenate([du, dt]).astype(np.float32)
    grid = np.stack(np.meshgrid(oefs, dbvs, indexing="ij"), -1).reshape(-1, 2)[perm]   # OEF is the slow axis
    close(y[:, :2], grid, rtol=1e-6)
    close(y[:, 2], 301.74327499379774 * y[:, 0] * y[:, 1], rtol=2e-6)                   # R2' = dw * dbv
    close(oracle32.signal_fwd(y[:, :2]), x, rtol=2e-5)
    assert dt.min() >= float(params["dbv_start"]) and dt.max() <= float(params["dbv_end"])


# -- model.py / logit_mvn.py --------------------------------------------------------------------------
def test_normalise_data(params):
    from oracle.oracle import Oracle
    x, single, multi = g("normalise", "x", "single", "multi")
    close(Oracle("f32", params).normalise(x), single.reshape(x.shape))
    close(Oracle("f32", params, multi_image_normalisation=True).normalise(x), multi.reshape(x.shape))


def test_reparam_and_moments(oracle32):
    q, z, y = g("reparam", "q", "z", "oef_dbv")
    close(oracle32.reparam(q, z), y)
    q, z, means, var = g("moments", "q", "z", "means", "variances")
    assert z.shape[1:] == (20, 2)
    m, v = oracle32.moments(q, z)
    close(m, means, rtol=2e-5)
    close(v, var, rtol=2e-4, atol=1e-9)


def test_log_probabilities(oracle32):
    obs, q, mvg, lm_mvg, diag = g("logprob", "obs", "q", "mvg", "lm_mvg", "diag")
    np.testing.assert_array_equal(mvg, lm_mvg)            # logit_mvn.py is the same arithmetic as model.py
    close(oracle32.logit_mvn_nlogp(obs, q), mvg, rtol=2e-5, atol=2e-5)
    close(oracle32.logit_gaussian_nlogp(obs[6:], q[6:]), diag, rtol=2e-5, atol=2e-5)


def test_synthetic_data_loss(oracle32):
    y, q, hyper = g("synth_loss", "y_true", "q", "hyper")
    close(oracle32.synthetic_data_loss(y, q), g("synth_loss", "mvg"), rtol=2e-5)
    close(oracle32.synthetic_data_loss(y, q, 2.0, 0.5), g("synth_loss", "mvg_ig"), rtol=2e-5)
    nl = oracle32.logit_gaussian_nlogp(y[:, :2], q).astype(np.float64)
    close(nl.mean(), g("synth_loss", "diag"), rtol=2e-5)
    s_o, s_d = 3 * np.tanh(q[:, 1].astype(np.float64)) - 1, 3 * np.tanh(q[:, 3].astype(np.float64)) - 1
    ig = stats.invgamma(2.0, scale=0.5)
    close((nl - ig.logpdf(np.exp(2 * s_o)) - ig.logpdf(np.exp(2 * s_d))).mean(), g("synth_loss", "diag_ig"), rtol=2e-5)
    ig_o, ig_d = stats.invgamma(hyper[0], scale=hyper[1]), stats.invgamma(hyper[2], scale=hyper[3])
    close((nl - ig_o.logpdf(np.exp(2 * s_o)) - ig_d.logpdf(np.exp(2 * s_d))).mean(), g("synth_loss", "diag_learned_ig"),
          rtol=2e-5)
    # use_r2p_loss (model.py:475-490): a normal fitted to the R2' of ten reparameterised draws
    z = g("synth_loss", "r2p_z")
    assert z.shape[1:] == (10, 2)
    m, v = (a.astype(np.float64) for a in oracle32.moments(q, z))
    r2p_nll = 0.5 * np.log(v[:, 2]) + 0.5 * (y[:, 2] - m[:, 2]) ** 2 / v[:, 2]
    close(oracle32.synthetic_data_loss(y, q) + r2p_nll.mean(), g("synth_loss", "mvg_r2p"), rtol=1e-4)


@pytest.mark.parametrize("case,kw", [("gaussian", {}), ("student_t5", dict(student_t_df=5)),
                                     ("log_data", dict(predict_log_data=True)),
                                     ("multi_image", dict(multi_image_normalisation=True)),
                                     ("student_t5_log_multi", dict(student_t_df=5, predict_log_data=True,
                                                                   multi_image_normalisation=True))])
def test_fine_tune_loss_variants(params, case, kw):
    from oracle.oracle import Oracle
    o = Oracle("f32", params, **kw)
    data, mask, pred, sigma = g("nll/inputs", "data", "mask", "pred", "sigma")
    nll = o.nll(data, mask, pred, sigma)
    close(nll * mask, g(f"nll/{case}", "per_voxel"), rtol=3e-5, atol=1e-4)
    close((nll * mask).astype(np.float64).sum() / mask.sum(), g(f"nll/{case}", "mean"), rtol=2e-5)


def test_fine_tune_loss_tiled_batch_and_homoscedastic(params, oracle32):
    data, mask, pred, sigma = g("nll/inputs", "data", "mask", "pred", "sigma")
    preds, mean, rows = g("nll/three_samples", "preds", "mean", "per_row")
    assert preds.shape[1] == 3 and rows.shape == (3, len(mask))     # copy s of voxel v is row s * N + v
    got = np.stack([oracle32.nll(data, mask, preds[:, s], sigma) * mask for s in range(3)])
    close(got, rows, rtol=3e-5, atol=1e-4)
    close(got.astype(np.float64).sum() / (3 * mask.sum()), mean, rtol=2e-5)      # both sums run over the tiled batch
    s0 = float(g("nll/homoscedastic", "sigma"))
    nll = oracle32.nll(data, mask, pred, np.full_like(pred, s0)) * mask
    close(nll, g("nll/homoscedastic", "per_voxel"), rtol=3e-5, atol=1e-4)
    close(nll.astype(np.float64).sum() / mask.sum(), g("nll/homoscedastic", "mean"), rtol=2e-5)


def test_kl_terms(oracle32):
    q, prior, mask, z = g("kl/sampled", "q", "prior", "mask", "z")
    assert z.shape[1:] == (70, 2)
    kl = oracle32.kl_samples(q, prior, z) * (mask > 0)
    close(kl, g("kl/sampled", "per_voxel"), rtol=1e-4, atol=1e-4)
    close(kl.astype(np.float64).sum() / mask.sum(), g("kl/sampled", "mean"), rtol=1e-4)
    close(oracle32.kl_closed(q, prior), g("kl/sampled", "closed_form"), rtol=1e-4, atol=1e-4)
    kd = oracle32.kl_diag(q, prior) * (mask > 0)
    close(kd, g("kl/diag", "per_voxel"), rtol=1e-4, atol=1e-5)
    close(kd.astype(np.float64).sum() / mask.sum(), g("kl/diag", "mean"), rtol=1e-4)


def test_smoothness_loss(oracle32):
    q, mask, loss, diag_loss = g("smoothness", "q", "mask", "loss", "diag_loss")
    close(oracle32.smoothness_loss(q, mask), loss, rtol=2e-5)
    close(oracle32.smoothness_loss(q, mask), diag_loss, rtol=2e-5)     # the 4-parameter family has the same means


# -- create_encoder / build_fine_tuner ----------------------------------------------------------------
@pytest.mark.parametrize("case", ["encoder_relu", "encoder_shared_gate"])
def test_two_stream_encoder(oracle32, case):
    """Conv3D layers exported in creation order (model.py:181, per block :144, :152, :156, :164, then :196, :211):
    stream 1 = the shared 1x1x1 chain, stream 2 = gated residual blocks with 3x3x1 convolutions, shared final
    layer, sigma head on stream 2's features."""
    w = weights_of(case)
    x, o1, o2, sg = g(f"{case}/voxels", "x", "out1", "out2", "sigma")
    g1, g2, gs = oracle32.encoder_fwd(centre_taps(w), x)    # (N,1,1,1,T): 'same' padding leaves the centre tap
    close(g1, o1, rtol=2e-5, atol=2e-5)
    close(g2, o2, rtol=2e-5, atol=2e-5)
    close(gs, sg, rtol=5e-5)
    xc, c1, c2, cs = g(f"{case}/crops", "x", "out1", "out2", "sigma")
    s2, ssg = oracle32.encoder_fwd_spatial(w, xc)
    close(s2, c2, rtol=2e-5, atol=2e-5)
    close(ssg, cs, rtol=5e-5)
    p1, _, _ = oracle32.encoder_fwd(centre_taps(w), xc.reshape(-1, 11))   # stream 1 has no spatial context
    close(p1.reshape(c1.shape), c1, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("case,gelu", [("encoder_layer_norm_relu", False), ("encoder_layer_norm_gelu", True)])
def test_encoder_with_layer_norm_and_dropout(oracle32, case, gelu):
    """use_layer_norm=True, dropout_rate=0.2 (model.py:131-140): the reference text's add_normalizer puts Dropout (the
    identity outside fit) and GroupNormalization(groups = 1, axis = -1) in front of BOTH activations of the residual
    path -- statistics over a voxel's channels for (N,1,1,1,T) batches, over a whole crop and its channels for crops;
    stream 1 and the skip path carry no normalisation."""
    w = weights_of(case)
    ln = g(f"{case}/weights", "ln")
    oracle32.lib.qbo_set_activation_gelu(1 if gelu else 0)
    try:
        x, o1, o2, sg = g(f"{case}/voxels", "x", "out1", "out2", "sigma")
        n = x.shape[0]
        s2, ssg = oracle32.encoder_fwd_spatial(w, x.reshape(n, 1, 1, 1, 11), ln=ln)
        close(s2.reshape(n, 5), o2, rtol=3e-5, atol=3e-5)
        close(ssg.reshape(n, 11), sg, rtol=1e-4)
        p1, _, _ = oracle32.encoder_fwd(centre_taps(w), x)
        close(p1, o1, rtol=2e-5, atol=2e-5)                                   # stream 1: no normalizer
        xc, c1, c2, cs = g(f"{case}/crops", "x", "out1", "out2", "sigma")
        s2, ssg = oracle32.encoder_fwd_spatial(w, xc, ln=ln)
        close(s2, c2, rtol=3e-5, atol=3e-5)
        close(ssg, cs, rtol=1e-4)
        plain, _ = oracle32.encoder_fwd_spatial(w, xc)
        assert np.max(np.abs(plain - c2)) > 1e-2                              # the layers are live in the fixture
    finally:
        oracle32.lib.qbo_set_activation_gelu(0)


def test_fine_tuner_elbo_from_the_reference_text(oracle32):
    """build_fine_tuner on a crop batch with S = 2 copies, then fine_tune_loss_fn + kl_loss on its outputs
    (train.py:315-320): the whole voxel-ELBO evaluation as the reference text computes it."""
    w = weights_of("fine_tuner")
    data, mask, prior, zs, zk = g("fine_tuner", "data", "mask", "prior", "zs", "zk")
    S = int(g("fine_tuner", "S"))
    B, X, Y, Z, T = data.shape
    n = B * X * Y * Z
    q, sigma = oracle32.encoder_fwd_spatial(w, data)
    pred, imgs = g("fine_tuner", "predictions", "predicted_images")
    assert pred.shape == (S * B, X, Y, Z, 5) and imgs.shape == (S * B, X, Y, Z, 2 * T)
    close(np.concatenate([q] * S), pred, rtol=2e-5, atol=2e-5)                 # copies concatenated along the batch
    close(np.concatenate([sigma] * S), imgs[..., T:], rtol=5e-5)
    p1, _, _ = oracle32.encoder_fwd(centre_taps(w), data.reshape(n, T))
    close(p1.reshape(prior.shape), prior, rtol=2e-5, atol=2e-5)                # prior = stream-1 output
    # draws: zs [S*B, X, Y, Z, 2] (copy s of crop b at row s*B + b), zk [K, S*B, X, Y, Z, 2]
    zs_v = zs.reshape(S, n, 2).transpose(1, 0, 2)
    y = oracle32.reparam(np.repeat(q.reshape(n, 5), S, 0), zs_v.reshape(n * S, 2))
    close(oracle32.signal_fwd(y).reshape(n, S, T).transpose(1, 0, 2).reshape(imgs[..., :T].shape), imgs[..., :T], rtol=3e-5)
    K = zk.shape[0]
    zk_v = zk.reshape(K, S, n, 2).transpose(2, 1, 0, 3).reshape(n, S * K, 2)
    out = oracle32.elbo(data.reshape(n, T), mask.reshape(n), q.reshape(n, 5), prior.reshape(n, 5), sigma.reshape(n, T),
                        zs_v, zk_v)
    nll_rows, kl_rows = g("fine_tuner", "nll_rows", "kl_rows")
    close(out["nll_v"] * mask.reshape(n), nll_rows.reshape(S, n).mean(0), rtol=5e-5, atol=1e-3)
    close(out["kl_v"] * (mask.reshape(n) > 0), kl_rows.reshape(S, n).mean(0), rtol=1e-4, atol=1e-4)
    close(out["nll"], g("fine_tuner", "nll"), rtol=1e-4)
    close(out["kl"], g("fine_tuner", "kl"), rtol=1e-4)
    close(out["elbo"], g("fine_tuner", "neg_elbo"), rtol=1e-4)                 # the north-star's ELBO tolerance
    close(oracle32.smoothness_loss(np.concatenate([q] * S), np.concatenate([mask] * S)), g("fine_tuner", "smoothness"),
          rtol=2e-5)


def test_population_prior_kl_diag(oracle32):
    """kl_loss with use_population_prior (diagonal family, model.py:687-716): every voxel against ONE prior carried
    in the predictions, plus the inverse-gamma cost on that prior times the batch axis."""
    q, prior, mask = g("kl/sampled", "q", "prior", "mask")
    pop = g("kl/population_diag", "pop_prior")
    n = len(q)
    pop5 = np.concatenate([np.broadcast_to(pop, (n, 4)), np.zeros((n, 1), np.float32)], -1)
    kd = oracle32.kl_diag(q, pop5) * (mask > 0)
    close(kd, g("kl/population_diag", "per_voxel"), rtol=1e-4, atol=1e-5)
    cost = oracle32.population_prior_cost(pop, n)        # the fixture's batch axis is the voxel axis: [N,1,1,1,8]
    close((kd.astype(np.float64).sum() + cost) / mask.sum(), g("kl/population_diag", "mean"), rtol=1e-4)
    assert abs(cost) > 1.0


def test_homoscedastic_fine_tuner(params, oracle32):
    """heteroscedastic_noise=False (model.py:277-281, 535-537): 'predicted_images' = [signal, one channel holding
    the exp-activated scalar], and fine_tune_loss_fn scores every tau with that one sigma."""
    w = weights_of("fine_tuner_homoscedastic")
    data, zs, imgs = g("fine_tuner_homoscedastic", "data", "zs", "predicted_images")
    s0 = float(g("fine_tuner_homoscedastic", "initial_im_sigma"))
    assert imgs.shape == (64, 12)
    close(imgs[:, 11], np.full(64, s0), rtol=1e-6)
    _, q, _ = oracle32.encoder_fwd(centre_taps(w), data)
    pred = oracle32.signal_fwd(oracle32.reparam(q, zs))
    close(pred, imgs[:, :11], rtol=3e-5)
    nll = oracle32.nll(data, np.ones(64, np.float32), pred, np.full_like(pred, s0))
    close(nll.astype(np.float64).mean(), g("fine_tuner_homoscedastic", "nll"), rtol=2e-5)


def test_gelu_encoder(oracle32):
    """activation_type='gelu' (the class default of EncoderTrainer, model.py:60; Keras' exact erf form) everywhere the
    reference applies its activation: the 1x1x1 layers (:119-120) and the two Activation layers of a block (:151, :155)
    -- where the second stream's input is activated AGAIN (gelu is not idempotent as relu is)."""
    w = weights_of("encoder_gelu")
    oracle32.set_activation("gelu")
    try:
        x, o1, o2, sg = g("encoder_gelu/voxels", "x", "out1", "out2", "sigma")
        g1, g2, gs = oracle32.encoder_fwd(centre_taps(w), x)
        close(g1, o1, rtol=2e-5, atol=2e-5)
        close(g2, o2, rtol=2e-5, atol=2e-5)
        close(gs, sg, rtol=5e-5)
        xc, c2, cs = g("encoder_gelu/crops", "x", "out2", "sigma")
        s2, ssg = oracle32.encoder_fwd_spatial(w, xc)
        close(s2, c2, rtol=2e-5, atol=2e-5)
        close(ssg, cs, rtol=5e-5)
    finally:
        oracle32.set_activation("relu")
    r1, _, _ = oracle32.encoder_fwd(centre_taps(w), x)
    assert np.abs(r1 - o1).max() > 1e-3          # relu on the same weights is a different network


def test_mixture_of_gaussians_kl(oracle32):
    """kl_loss with mog_components = 3 (model.py:666-685): -entropy + the MEAN of the components' Gaussian NLLs at one
    reparameterised draw per dimension."""
    q, mask = g("kl/sampled", "q", "mask")
    comps, z = g("kl/mog", "components", "z")
    assert comps.shape == (3, 4) and z.shape == (len(q), 2)
    kl = oracle32.kl_mog(q, comps, z) * (mask > 0)
    close(kl, g("kl/mog", "per_voxel"), rtol=1e-4, atol=1e-5)
    close(kl.astype(np.float64).sum() / mask.sum(), g("kl/mog", "mean"), rtol=1e-4)

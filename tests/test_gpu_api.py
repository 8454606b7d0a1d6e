"""GPU tests of the reference-shaped Python interface (qbold_vi_amd.signals / model / logit_mvn):
they read like the calls the reference's own scripts make, and check against the CPU oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


@pytest.fixture(scope="module")
def trainer(params):
    from qbold_vi_amd import EncoderTrainer
    return EncoderTrainer(system_params=params, no_units=60, use_layer_norm=False, dropout_rate=0.0,
                          no_intermediate_layers=2, student_t_df=200, initial_im_sigma=0.05,
                          activation_type='relu', multi_image_normalisation=False,
                          channelwise_gating=True, infer_inv_gamma=False, use_population_prior=False,
                          use_mvg=True, predict_log_data=False)


def oracle_weights(model, gate_offset):
    w = model.get_weights()
    w["gate_offset"] = gate_offset
    w["meta"] = dict(T=11, U=60, L=2, channelwise_gating=True, taps=9)
    return w


def gaussian_kl(q, p):
    """KL(q || p) of the bivariate Gaussians in logit space, float64."""
    def chol(v):
        v = v.astype(np.float64)
        so, sd = np.tanh(v[:, 1]) * 3 - 1, np.tanh(v[:, 3]) * 3 - 1
        c = np.tanh(v[:, 4]) * np.exp(-2.0)
        L = np.zeros((len(v), 2, 2))
        L[:, 0, 0], L[:, 1, 0], L[:, 1, 1] = np.exp(so), c, np.exp(sd)
        return np.stack([v[:, 0], v[:, 2]], -1), L, so + sd
    mq, Lq, ldq = chol(q)
    mp, Lp, ldp = chol(p)
    Sq = Lq @ Lq.transpose(0, 2, 1)
    Spi = np.linalg.inv(Lp @ Lp.transpose(0, 2, 1))
    d = (mp - mq)[:, :, None]
    tr = np.trace(Spi @ Sq, axis1=1, axis2=2)
    quad = (d.transpose(0, 2, 1) @ Spi @ d)[:, 0, 0]
    return 0.5 * (tr + quad - 2.0 + 2.0 * (ldp - ldq))


def test_signal_layer_self_check_point(params, oracle32):
    """The reference's only self-check (signals.py:307-314): the layer at [[[[[0.4, 0.12]]]]]."""
    from qbold_vi_amd import SignalGenerationLayer
    p = dict(params, simulate_noise='False')
    layer = SignalGenerationLayer(p, True, True)
    out = layer(dev(np.array([[[[[0.4, 0.12]]]]], np.float32)))
    assert out.shape == (1, 1, 1, 1, 11)
    want = oracle32.signal_fwd(np.array([[0.4, 0.12]], np.float32))[0]
    np.testing.assert_allclose(out.cpu().numpy().ravel(), want, rtol=1e-5)
    with pytest.raises(AssertionError):
        layer(dev(np.zeros((4, 3), np.float32)))
    # gradient entry: finite differences of the layer itself
    y = dev(np.array([[0.4, 0.12], [0.3, 0.03]], np.float32))
    g = torch.ones((2, 11), device="cuda")
    got = layer.gradient(y, g).cpu().numpy()
    eps = 1e-3
    for k in range(2):
        d = torch.zeros_like(y)
        d[:, k] = eps
        fd = ((layer(y + d) - layer(y - d)).sum(-1) / (2 * eps)).cpu().numpy()
        np.testing.assert_allclose(got[:, k], fd, rtol=2e-2, atol=2e-3)


def test_noise_model_statistics(params):
    from qbold_vi_amd import SignalGenerationLayer
    clean = SignalGenerationLayer(dict(params, simulate_noise='False'), True, True)
    noisy = SignalGenerationLayer(dict(params, simulate_noise='True'), True, True)
    y = dev(np.tile(np.array([[0.4, 0.025]], np.float32), (200000, 1)))
    c, n = clean(y), noisy(y)
    resid = (n - c) / c.mean(0, keepdim=True)
    # std = 1/(U(50,120) * norm_snr): E[1/snr^2] over the uniform = 1/(50*120)
    norm_snr = np.array([0.985, 1.00, 1.01, 1., 0.97, 0.95, 0.93, 0.90, 0.86, 0.83, 0.79])
    want = np.sqrt(1.0 / 6000.0) / norm_snr
    np.testing.assert_allclose(resid.std(0).cpu().numpy(), want, rtol=2e-2)
    assert abs(float(resid.mean())) < 1e-4
    assert not torch.equal(noisy(y), n)  # fresh noise per call


def test_create_synthetic_dataset(params, oracle32):
    from qbold_vi_amd import create_synthetic_dataset
    p = dict(params, simulate_noise='False')
    x, y = create_synthetic_dataset(p, True, True, 0.0, uniform_prop=0.1, sample_size=50)
    assert x.shape == (2500, 11) and y.shape == (2500, 3)
    yn = y.cpu().numpy()
    assert yn[:, 0].min() >= 0.05 - 1e-6 and yn[:, 0].max() <= 0.8 + 1e-6
    assert yn[:, 1].min() >= 0.003 - 1e-6 and yn[:, 1].max() <= 0.195 + 1e-6
    np.testing.assert_allclose(x.cpu().numpy(), oracle32.signal_fwd(yn[:, :2]), rtol=1e-5)
    np.testing.assert_allclose(yn[:, 2], 301.74327499379774 * yn[:, 0] * yn[:, 1], rtol=1e-5)
    # meshgrid of 50 x 50 values (clipped OEF draws may coincide at the bounds)
    assert 40 <= len(np.unique(yn[:, 0])) <= 50 and len(np.unique(yn[:, 1])) == 50


def test_encoder_trainer_interface(trainer, params, oracle32):
    from oracle.oracle import synth_inputs
    from qbold_vi_amd import SignalGenerationLayer
    model, _ = trainer.create_encoder(gate_offset=-3.0, resid_init_std=0.05, no_ip_images=11)
    w = oracle_weights(model, -3.0)
    n = 1000
    x, truth = synth_inputs(n, params, seed=3, oracle=oracle32)
    x5 = dev(x).reshape(n, 1, 1, 1, 11)
    out1, out2, sigma = model(x5)
    assert out1.shape == (n, 1, 1, 1, 5) and sigma.shape == (n, 1, 1, 1, 11)
    w1, w2, ws = oracle32.encoder_fwd(w, x)
    assert np.abs(out1.reshape(n, 5).cpu().numpy() - w1).max() < 2e-5
    assert np.abs(out2.reshape(n, 5).cpu().numpy() - w2).max() < 2e-5
    np.testing.assert_allclose(trainer.normalise_data(dev(x)).cpu().numpy(), oracle32.normalise(x),
                               rtol=1e-5, atol=1e-6)
    # pre-training loss, model.py:449-514
    y3 = np.concatenate([truth, truth[:, :1] * truth[:, 1:2]], -1).astype(np.float32)
    got = float(trainer.synthetic_data_loss(dev(y3).reshape(n, 1, 1, 1, 3), out1))
    assert abs(got - oracle32.synthetic_data_loss(y3, w1)) < 1e-4 * abs(got) + 1e-5
    # transforms
    v = dev(np.linspace(-3, 3, 64).astype(np.float32))
    np.testing.assert_allclose(trainer.transform_std(v).cpu().numpy(), np.tanh(v.cpu().numpy()) * 3 - 1,
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(trainer.inv_transform_std(trainer.transform_std(v * 0.3)).cpu().numpy(),
                               (v * 0.3).cpu().numpy(), rtol=1e-4, atol=1e-5)
    lg = dev(np.random.default_rng(0).normal(size=(32, 2)).astype(np.float32))
    fw = trainer.forward_transform(lg)
    np.testing.assert_allclose(trainer.backwards_transform(fw, True).cpu().numpy(), lg.cpu().numpy(),
                               rtol=1e-4, atol=1e-5)
    # weights round trip
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "pt_model.npz")
        model.save_weights(path)
        model2, _ = trainer.create_encoder(gate_offset=-3.0, resid_init_std=0.3, no_ip_images=11)
        model2.set_weights({k: v * 0 for k, v in model2.get_weights().items()})
        model2.load_weights(path)
        assert torch.equal(model2(x5)[1], out2)

    # fine tuner: unfused outputs + losses, and the fused ELBO
    sig_layer = SignalGenerationLayer(dict(params, simulate_noise='False'), True, True)
    full_model = trainer.build_fine_tuner(model, sig_layer, None, None)
    rng = np.random.default_rng(4)
    mask = (rng.uniform(size=(n, 1)) > 0.3).astype(np.float32)
    mask5 = dev(mask).reshape(n, 1, 1, 1, 1)
    outs = full_model([x5, mask5])
    assert outs['predictions'].shape == (n, 1, 1, 1, 5)
    assert outs['predicted_images'].shape == (n, 1, 1, 1, 22)
    y_true = torch.cat([x5, mask5], -1)
    nll = trainer.fine_tune_loss_fn(y_true, outs['predicted_images'])
    pi = outs['predicted_images'].reshape(n, 22).cpu().numpy()
    want = oracle32.nll(x, mask[:, 0], pi[:, :11], pi[:, 11:])
    assert abs(float(nll) - (want * mask[:, 0]).sum() / mask.sum()) < 1e-4 * abs(float(nll))
    prior_mask = torch.cat([out1, mask5], -1)
    kl = trainer.kl_loss(prior_mask, outs['predictions'], no_samples=70, seed=5)
    zk = oracle32.philox_normals(5, 1, 0, n, 70)
    klv = oracle32.kl_samples(w2, w1, zk)
    want_kl = np.where(mask[:, 0] > 0, klv, 0).sum() / mask.sum()
    assert abs(float(kl) - want_kl) < 1e-4 * abs(want_kl) + 1e-5
    # mvg_kl restates the reference's closed form (model.py:612-652) term by term ...
    closed = trainer.mvg_kl(prior_mask, outs['predictions']).reshape(-1).cpu().numpy()
    np.testing.assert_allclose(closed, oracle32.kl_closed(w2, w1), rtol=1e-3, atol=1e-4)
    # ... which builds Sigma_p^-1 as L^-1 L^-T instead of L^-T L^-1 and is therefore exact only for
    # zero covariance (it is never called by the reference's training).  The K -> infinity limit
    # of the sampled KL is the true Gaussian KL in logit space, computed here in float64:
    big = trainer.mvg_kl_samples(prior_mask, outs['predictions'], no_samples=4000).reshape(-1).cpu().numpy()
    assert abs(big.mean() - gaussian_kl(w2, w1).mean()) < 0.02 * abs(gaussian_kl(w2, w1).mean()) + 1e-3
    assert float(trainer.smoothness_loss(prior_mask, outs['predictions'])) == 0.0
    fused = full_model.elbo(x5, mask5, out1, no_samples=1, kl_samples=70, seed=9)
    zs = oracle32.philox_normals(9, 0, 0, n, 1)
    zk = oracle32.philox_normals(9, 1, 0, n, 70)
    ref = oracle32.elbo(x, mask[:, 0], w2, w1, ws, zs, zk)
    assert abs(float(fused['elbo']) - ref['elbo']) < 1e-4 * abs(ref['elbo'])
    # no_samples = S > 1: the reference draws kl_samples KL samples per copy of its S-fold tiled batch, i.e.
    # S * kl_samples per voxel (default, kl_tiled=True); kl_tiled=False draws kl_samples per voxel -- same KL in
    # expectation
    S3 = 3
    tiled = full_model.elbo(x5, mask5, out1, no_samples=S3, kl_samples=70, seed=9)
    ref3 = oracle32.elbo(x, mask[:, 0], w2, w1, ws, oracle32.philox_normals(9, 0, 0, n, S3),
                         oracle32.philox_normals(9, 1, 0, n, S3 * 70))
    assert abs(float(tiled['kl']) - ref3['kl']) < 1e-4 * abs(ref3['kl']) + 1e-5
    assert abs(float(tiled['elbo']) - ref3['elbo']) < 1e-4 * abs(ref3['elbo'])
    flat = full_model.elbo(x5, mask5, out1, no_samples=S3, kl_samples=70, seed=9, kl_tiled=False)
    d = (flat['nll_kl'][:, 1] - tiled['nll_kl'][:, 1]).cpu().numpy()
    assert abs(float(flat['kl']) - float(tiled['kl'])) < 6.0 * d.std() / np.sqrt(n) + 1e-6
    assert trainer.kl_draws(70, 4) == 280 and trainer.kl_draws(70, 4, kl_tiled=False) == 70
    # posterior moments via the trainer
    means, var = trainer.calculate_means(out2, None, include_r2p=True, return_stds=True,
                                         no_samples=200, seed=3)
    zm = oracle32.philox_normals(3, 2, 0, n, 200)
    wm, wv = oracle32.moments(w2, zm)
    assert np.abs(means.reshape(n, 3).cpu().numpy()[:, :2] - wm[:, :2]).max() < 1e-5
    np.testing.assert_allclose(var.reshape(n, 3).cpu().numpy(), wv, rtol=1e-3, atol=1e-9)
    samples = trainer.create_samples(out2, None, 7)
    assert samples.shape == (n, 1, 1, 1, 2, 7)
    assert float(trainer.oef_metric(dev(y3).reshape(n, 1, 1, 1, 3), out1)) >= 0.0


def test_disabled_branches_fail_loudly(params):
    from qbold_vi_amd import EncoderTrainer, SignalGenerationLayer
    # the pairs the reference itself cannot run (shape errors in its own code) and what is left unbuilt
    with pytest.raises(NotImplementedError, match="shape error in the reference"):
        EncoderTrainer(params, use_population_prior=True, use_mvg=True, activation_type='relu')
    with pytest.raises(NotImplementedError, match="shape error in the reference"):
        EncoderTrainer(params, use_population_prior=False, use_mvg=True, infer_inv_gamma=True, activation_type='relu')
    with pytest.raises(NotImplementedError, match="activation_type"):
        EncoderTrainer(params, use_population_prior=False, activation_type='selu')
    # built since round 4: dropout and GroupNormalization (tests/test_gpu_normalizer.py)
    EncoderTrainer(params, use_population_prior=False, activation_type='relu', dropout_rate=0.1, use_layer_norm=True)
    # built since round 3: gelu, the homoscedastic sigma, the population prior with the diagonal family
    EncoderTrainer(params, use_population_prior=False, activation_type='gelu')
    EncoderTrainer(params, use_population_prior=False, activation_type='relu', heteroscedastic_noise=False)
    EncoderTrainer(params, use_population_prior=True, use_mvg=False, mog_components=1, activation_type='relu')
    EncoderTrainer(params, use_population_prior=True, use_mvg=False, mog_components=3, activation_type='relu')
    # misalignment / variable_hct are built (test_signal_variable_hct_and_misalignment); their
    # gradient is not -- they only occur in synthetic-data generation (signals.py:251-300)
    layer = SignalGenerationLayer(dict(params, simulate_noise='False'), True, True, misaligned_prob=0.1)
    with pytest.raises(NotImplementedError):
        layer.gradient(torch.zeros(4, 2, device="cuda"), torch.zeros(4, 11, device="cuda"))


def test_logit_mvn(trainer, oracle32):
    from qbold_vi_amd import LogitMVN
    mvn = LogitMVN(trainer.context)
    rng = np.random.default_rng(1)
    p = rng.normal(size=(500, 5)).astype(np.float32)
    y = np.stack([rng.uniform(0.05, 0.8, 500), rng.uniform(0.003, 0.195, 500)], -1).astype(np.float32)
    got = mvn.logit_gaussian_mvg_log_prob(dev(y), dev(p)).cpu().numpy()
    ref = oracle32.logit_mvn_nlogp(y, p)
    assert np.max(np.abs(got - ref) / (np.abs(ref) + 1.0)) < 1e-4


def test_fit_wls_matches_loglinear_restatement(oracle32, params, tmp_path, monkeypatch):
    """loglinear.fit_wls (loglinear.py:68-105) on image-shaped data, as its __main__ calls it."""
    import os
    from oracle.oracle import fit_wls as fit_wls_ref
    from qbold_vi_amd import loglinear, nifti
    rng = np.random.default_rng(8)
    shape = (3, 12, 10, 8)
    n = int(np.prod(shape))
    y = np.stack([rng.uniform(0.1, 0.75, n), rng.uniform(0.005, 0.15, n)], -1)
    sig = (oracle32.signal_fwd(y) * rng.uniform(100, 500, (n, 1)) * (1 + 0.02 * rng.normal(size=(n, 11)))).astype(np.float32)
    sig[5, 8] = 0.0       # ln -> -inf -> 0 (loglinear.py:70-71)
    sig[6, 6] = -3.0      # ln -> nan -> 0
    sig[7] = 0.0          # whole voxel empty: 0/0 -> NaN survives np.clip
    sig = sig.reshape(shape + (11,))
    monkeypatch.chdir(os.path.join(os.path.dirname(__file__), ".."))   # fit_wls reads ./config like the reference
    oef, dbv, r2p = loglinear.fit_wls(sig)
    assert oef.shape == dbv.shape == r2p.shape == shape + (1,)
    ro, rd, rr = fit_wls_ref(sig, params)
    o, d, r = (t.cpu().numpy().astype(np.float64) for t in (oef, dbv, r2p))
    np.testing.assert_allclose(r, rr, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(d, rd, rtol=1e-4, atol=2e-6)
    assert np.isnan(o.reshape(-1)[7]) and np.isnan(ro.reshape(-1)[7])
    ok = ~np.isnan(ro)
    # OEF divides by DBV: compare where the fit is not within rounding of a clip edge or a pole
    stable = ok & (np.abs(rd) > 4e-3)
    np.testing.assert_allclose(o[stable], ro[stable], rtol=2e-3, atol=1e-5)
    assert (np.abs(o[ok] - ro[ok]) > 1e-3).mean() < 0.01
    assert 0.2 < (ro[ok] < 0.8).mean()                       # a real mix of clipped / unclipped voxels
    loglinear.save_predictions([oef, dbv, r2p], str(tmp_path / "wls"))
    im, _ = nifti.load(str(tmp_path / "wls_dbv.nii.gz"))
    assert im.shape == (12, 10, 8, 3)
    np.testing.assert_array_equal(im[..., 1], dbv[1, ..., 0].cpu().numpy())
    # a tau grid without tau = 0 is an error, as the reference's empty s0_id index would be
    from qbold_vi_amd.ops import Context
    bad = Context(dict(params, tau_start="-0.015"), True, True)
    with pytest.raises(RuntimeError, match="no tau equals 0"):
        bad.wls_fit(torch.ones(4, bad.T, device="cuda"))


def test_diagonal_family_use_mvg_false(params, oracle32):
    """use_mvg=False is the reference's argparse default (train.py:181): 4-parameter predictions,
    independent draws, closed-form KL (model.py:33-37, 191-193, 406-421, 686-721)."""
    from oracle.oracle import Oracle, synth_inputs
    from qbold_vi_amd import EncoderTrainer, SignalGenerationLayer
    tr = EncoderTrainer(system_params=params, no_units=30, use_layer_norm=False, dropout_rate=0.0,
                        no_intermediate_layers=1, student_t_df=2, initial_im_sigma=0.08, activation_type='relu',
                        multi_image_normalisation=True, channelwise_gating=True, infer_inv_gamma=False,
                        use_population_prior=False, use_mvg=False, predict_log_data=True, no_samples=2)
    orc = Oracle("f32", params, multi_image_normalisation=True, predict_log_data=True, student_t_df=2)
    model, _ = tr.create_encoder(gate_offset=0.0, resid_init_std=0.1, no_ip_images=11)
    w = model.get_weights()
    assert w["Wf"].shape == (30, 4) and w["bf"].shape == (4,)
    model.set_weights(w)                                   # round trip through the 4-column form
    n = 600
    x, y = synth_inputs(n, seed=3, oracle=oracle32)
    xd = dev(x).reshape(n, 1, 1, 1, 11)
    o1, o2, sg = model.predict(xd)
    assert o1.shape == o2.shape == (n, 1, 1, 1, 4)
    w5 = dict(w, Wf=np.concatenate([w["Wf"], np.zeros((30, 1), np.float32)], 1), bf=np.concatenate([w["bf"], [0.0]]).astype(np.float32),
              gate_offset=0.0, meta=dict(T=11, U=30, L=1, channelwise_gating=True, taps=9))
    w1, w2, wsg = orc.encoder_fwd(w5, x)
    assert np.abs(o2.reshape(n, 4).cpu().numpy() - w2[:, :4]).max() < 2e-5 and np.abs(w2[:, 4]).max() == 0.0
    # pre-training loss: logit_gaussian_log_prob
    y3 = np.concatenate([y, y[:, :1]], -1).astype(np.float32)
    got = float(tr.synthetic_data_loss(dev(y3).reshape(n, 1, 1, 1, 3), o1))
    want = float(orc.logit_gaussian_nlogp(y, w1).mean())
    assert abs(got - want) < 1e-4 * abs(want) + 1e-5
    lp = tr.logit_gaussian_log_prob(dev(y), o1)
    np.testing.assert_allclose(lp.reshape(-1).cpu().numpy(), orc.logit_gaussian_nlogp(y, w1), rtol=1e-4, atol=1e-4)
    # closed-form KL through kl_loss: true = [prior (4), mask]
    mask = (np.random.default_rng(1).uniform(size=n) > 0.2).astype(np.float32)
    true = torch.cat([o1, dev(mask).reshape(n, 1, 1, 1, 1)], -1)
    q_t = torch.cat([o2, o2], 0)                             # no_samples = 2: the S-fold tiled distribution
    kl = float(tr.kl_loss(true, q_t))
    klv = orc.kl_diag(w2, w1)
    assert abs(kl - float(np.where(mask > 0, klv, 0).sum() / mask.sum())) < 1e-4 * abs(kl) + 1e-6
    # ELBO: sampled NLL (Student-t, log data, 3-image normalisation) + closed-form KL
    full = tr.build_fine_tuner(model, SignalGenerationLayer(dict(params, simulate_noise='False'), True, True))
    out = full.elbo(xd, dev(mask).reshape(n, 1, 1, 1, 1), o1, no_samples=4, seed=5)
    zs = orc.philox_normals(5, 0, 0, n, 4)
    e = orc.elbo(x, mask, w2, w1, wsg, zs, np.zeros((n, 1, 2), np.float32))
    assert abs(float(out["nll"]) - e["sums"][0] / e["sums"][2]) < 1e-4 * abs(float(out["nll"]))
    assert abs(float(out["kl"]) - kl) < 1e-5 * abs(kl) + 1e-7
    assert out["q"].shape == (n, 4)
    # sampling: independent normals per dimension
    s = tr.create_samples(o2, None, 5)
    assert s.shape == (n, 1, 1, 1, 2, 5)
    means = tr.calculate_means(o2, None, include_r2p=True)
    assert means.shape == (n, 1, 1, 1, 3) and bool(torch.isfinite(means).all())


def test_kl_diag_gradient(params, oracle32):
    from oracle.oracle import Oracle
    from qbold_vi_amd.ops import Context
    ctx = Context(params, True, True)
    o64 = Oracle("f64", params)
    rng = np.random.default_rng(2)
    n = 300
    q = (rng.normal(size=(n, 5)) * 0.5).astype(np.float32)
    p = (rng.normal(size=(n, 5)) * 0.5).astype(np.float32)
    mask = (rng.uniform(size=n) > 0.3).astype(np.float32)
    g = torch.full((n, 5), 0.25, device="cuda")
    sums, kl = ctx.kl_diag(dev(q), dev(p), dev(mask), g_q=g)
    want = oracle32.kl_diag(q, p)
    np.testing.assert_allclose(kl.cpu().numpy(), want, rtol=2e-5, atol=1e-6)
    assert abs(float(sums[1]) - float(want[mask > 0].astype(np.float64).sum())) < 1e-5 * float(sums[1])
    assert float(sums[2]) == float(mask.sum()) and float(sums[0]) == 0.0
    g = g.cpu().numpy() - 0.25                                # the gradient is ADDED
    assert np.abs(g[mask == 0]).max() == 0.0 and np.abs(g[:, 4]).max() == 0.0
    q64 = q.astype(np.float64)
    for k in range(4):
        d = np.zeros_like(q64); d[:, k] = 1e-6
        fd = (o64.kl_diag(q64 + d, p) - o64.kl_diag(q64 - d, p)) / 2e-6
        np.testing.assert_allclose(g[mask > 0, k], fd[mask > 0], rtol=2e-4, atol=2e-5)

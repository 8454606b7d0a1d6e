"""The HIP kernels (through the reference-shaped host API and the C ABI) held to fixtures produced by the
REFERENCE'S OWN SOURCE TEXT (tests/golden/reference_text_goldens.npz; see tests/test_reference_text.py and
tests/golden/make_reference_text_goldens.py for what these fixtures are and are not: wiring, not TensorFlow's
arithmetic).  Random draws are the ones the reference code consumed, passed to the kernels explicitly."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_text_goldens.npz"))


def g(case, *names):
    out = [G[f"{case}/{n}"] for n in names]
    return out[0] if len(out) == 1 else out


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))      # e.g. logit(0) = -inf on both sides
    a, b = np.where(same_inf, 0.0, a), np.where(same_inf, 0.0, b)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert np.all(err <= 0), f"max excess {err.max():.3e}; max |diff| {np.abs(a - b).max():.3e}"


def trainer(params, **kw):
    from qbold_vi_amd.model import EncoderTrainer
    base = dict(no_intermediate_layers=2, no_units=12, activation_type="relu", student_t_df=200,
                initial_im_sigma=0.05, multi_image_normalisation=False, channelwise_gating=True, use_mvg=True,
                use_population_prior=False, no_samples=1, heteroscedastic_noise=True, predict_log_data=False)
    base.update(kw)
    return EncoderTrainer(params, **base)


def encoder_of(params, case, **kw):
    """An EncoderModel carrying the weights the reference's create_encoder built (creation-order export)."""
    from oracle.oracle import WEIGHT_NAMES
    U, L = int(G[f"{case}/weights/U"]), int(G[f"{case}/weights/L"])
    tr = trainer(params, no_units=U, no_intermediate_layers=L,
                 channelwise_gating=bool(G[f"{case}/weights/channelwise_gating"]), **kw)
    model, _ = tr.create_encoder(gate_offset=float(G[f"{case}/weights/gate_offset"]), resid_init_std=0.05, no_ip_images=11)
    model.set_weights({n: G[f"{case}/weights/{n}"] for n in WEIGHT_NAMES})
    return tr, model


# -- signals.py ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,full,blood", [("full_blood", True, True), ("full_noblood", True, False),
                                             ("loglinear_blood", False, True), ("loglinear_noblood", False, False)])
def test_forward_model_variants(params, case, full, blood):
    from qbold_vi_amd.signals import SignalGenerationLayer
    lay = SignalGenerationLayer(dict(params, simulate_noise="False"), full, blood)
    y, s, taus = g(f"signal/{case}", "oef_dbv", "signal", "taus")
    np.testing.assert_array_equal(lay.context.taus, taus)
    close(lay(dev(y)), s, rtol=2e-5)                          # table mode (the default of the product)
    lay.context.set_tissue_mode("literal")
    close(lay(dev(y)), s, rtol=2e-5)                          # 129-node Simpson sum in-kernel


def test_forward_model_shapes_protocol_hct_misalignment(params):
    from qbold_vi_amd.signals import SignalGenerationLayer
    P = dict(params, simulate_noise="False")
    lay = SignalGenerationLayer(P, True, True)
    y5, s5 = g("signal/shape5d", "oef_dbv", "signal")
    out = lay(dev(y5))
    assert tuple(out.shape) == s5.shape
    close(out, s5, rtol=2e-5)
    y, dw, r2p = g("signal/dw_r2p", "oef_dbv", "dw", "r2p")
    close(lay.calculate_dw(dev(y[:, 0]), lay.hct), dw, rtol=1e-6)
    close(lay.calculate_r2p(dev(y[:, 0]), dev(y[:, 1]), lay.hct), r2p, rtol=1e-6)
    lay24 = SignalGenerationLayer(dict(P, tau_start="-0.028", tau_end="0.065", tau_step="0.004"), True, True)
    y, s, taus = g("signal/protocol24", "oef_dbv", "signal", "taus")
    np.testing.assert_array_equal(lay24.context.taus, taus)
    close(lay24(dev(y)), s, rtol=2e-5)
    yh, sh = g("signal/variable_hct", "oef_dbv_hct", "signal")
    close(SignalGenerationLayer(P, True, True, variable_hct=True)(dev(yh)), sh, rtol=2e-5)
    # misalignment with the reference's draws made explicit (signals.py:80-96)
    y, s, u, idx, zo, zd, prob = g("signal/misaligned", "oef_dbv", "signal", "u_misaligned", "from_index", "z_oef",
                                    "z_dbv", "prob")
    alt = np.stack([np.clip(zo[:, 0] * np.float32(0.15) + y[:, 0], 0.05, 0.8),
                    np.clip(zd[:, 0] * np.float32(0.05) + y[:, 1], 0.002, 0.3)], -1).astype(np.float32)
    from_idx = np.where(u[:, 0] < prob, idx[:, 0], 11).astype(np.int32)
    close(lay.context.signal_fwd_ex(dev(y), None, dev(alt), dev(from_idx)), s, rtol=2e-5)


# -- model.py / logit_mvn.py --------------------------------------------------------------------------
def test_transforms_and_normalise(params):
    from qbold_vi_amd.logit_mvn import LogitMVN
    tr = trainer(params)
    lm = LogitMVN(tr.context)
    raw, y = g("transforms", "raw", "oef_dbv")
    assert tr._se_idx == int(g("transforms", "se_idx"))
    for obj, pre in ((tr, ""), (lm, "lm_")):
        close(obj.transform_std(dev(raw)), g("transforms", pre + "transform_std"), rtol=2e-6)
        close(obj.transform_offdiag(dev(raw)), g("transforms", pre + "transform_offdiag"), rtol=2e-6, atol=1e-8)
        close(obj.forward_transform(dev(raw)), g("transforms", pre + "forward_transform"), rtol=2e-6)
        close(obj.backwards_transform(dev(y), True), g("transforms", pre + "backwards_transform_logit"), rtol=2e-5, atol=2e-6)
    close(tr.backwards_transform(dev(y), False), g("transforms", "backwards_transform"), rtol=2e-6, atol=1e-7)
    close(tr.inv_transform_std(tr.transform_std(dev(raw))), g("transforms", "inv_transform_std"), rtol=1e-3, atol=1e-4)
    x, single, multi = g("normalise", "x", "single", "multi")
    close(tr.normalise_data(dev(x.reshape(-1, 1, 1, 1, 11))), single)
    close(trainer(params, multi_image_normalisation=True).normalise_data(dev(x.reshape(-1, 1, 1, 1, 11))), multi)


def test_reparam_moments_and_log_probabilities(params):
    from qbold_vi_amd.logit_mvn import LogitMVN
    from qbold_vi_amd.model import ReparamTrickLayer
    tr = trainer(params)
    q, z, y = g("reparam", "q", "z", "oef_dbv")
    n = len(q)
    smp = ReparamTrickLayer(tr)([dev(q.reshape(n, 1, 1, 1, 5)), None], z=dev(z))
    assert tuple(smp.shape) == (n, 1, 1, 1, 2)
    close(smp.reshape(n, 2), y)
    q, z, means, var = g("moments", "q", "z", "means", "variances")
    m, v = tr.context.posterior_moments(dev(q), 20, z=dev(z))
    close(m, means, rtol=2e-5)
    close(v, var, rtol=2e-4, atol=1e-9)
    obs, q, mvg, diag = g("logprob", "obs", "q", "mvg", "diag")
    close(tr.logit_gaussian_mvg_log_prob(dev(obs), dev(q.reshape(n, 1, 1, 1, 5))).reshape(n), mvg, rtol=2e-5, atol=2e-5)
    close(LogitMVN(tr.context).logit_gaussian_mvg_log_prob(dev(obs), dev(q.reshape(n, 1, 1, 1, 5))).reshape(n), mvg,
          rtol=2e-5, atol=2e-5)
    td = trainer(params, use_mvg=False)
    close(td.logit_gaussian_log_prob(dev(obs[6:]), dev(q[6:, :4].reshape(n - 6, 1, 1, 1, 4))).reshape(n - 6), diag,
          rtol=2e-5, atol=2e-5)
    swr_obs, swr, lm_swr, lcd = g("logprob", "swr_obs", "swr", "lm_swr", "log_chol_det")
    so, sd, cov = tr.transform_std(dev(q[:, 1])), tr.transform_std(dev(q[:, 3])), tr.transform_offdiag(dev(q[:, 4]))
    close(LogitMVN.squared_whitened_residual(dev(swr_obs), dev(q[:, [0, 2]]), so, sd, cov), lm_swr, rtol=2e-5, atol=1e-5)
    close(LogitMVN.calculate_log_chol_det(so, sd), lcd, rtol=2e-6, atol=1e-6)


def test_synthetic_data_loss(params):
    y, q, hyper = g("synth_loss", "y_true", "q", "hyper")
    n = len(q)
    y5, q5 = dev(y.reshape(n, 1, 1, 1, 3)), dev(q.reshape(n, 1, 1, 1, 5))
    tr, td = trainer(params), trainer(params, use_mvg=False)
    close(tr.synthetic_data_loss(y5, q5, False, 0.0, 0.0), g("synth_loss", "mvg"), rtol=2e-5)
    close(tr.synthetic_data_loss(y5, q5, False, 2.0, 0.5), g("synth_loss", "mvg_ig"), rtol=2e-5)
    close(td.synthetic_data_loss(y5, q5[..., :4], False, 0.0, 0.0), g("synth_loss", "diag"), rtol=2e-5)
    close(td.synthetic_data_loss(y5, q5[..., :4], False, 2.0, 0.5), g("synth_loss", "diag_ig"), rtol=2e-5)
    tl = trainer(params, use_mvg=False, infer_inv_gamma=True)
    q8 = torch.cat([q5[..., :4], dev(hyper).expand(n, 1, 1, 1, 4)], -1)
    close(tl.synthetic_data_loss(y5, q8, False, 0.0, 0.0), g("synth_loss", "diag_learned_ig"), rtol=2e-5)
    # use_r2p_loss with the reference's ten draws per voxel
    z = g("synth_loss", "r2p_z")
    lv = tr.context.logit_mvn_nlogp(dev(y[:, :2]), dev(q)).clone()
    tr.context.r2p_loss_bwd(dev(y), dev(q), 10, z=dev(z), loss_v=lv, want_grad=False)
    close(lv.mean(), g("synth_loss", "mvg_r2p"), rtol=1e-4)


@pytest.mark.parametrize("case,kw", [("gaussian", {}), ("student_t5", dict(student_t_df=5)),
                                     ("log_data", dict(predict_log_data=True)),
                                     ("multi_image", dict(multi_image_normalisation=True)),
                                     ("student_t5_log_multi", dict(student_t_df=5, predict_log_data=True,
                                                                   multi_image_normalisation=True))])
def test_fine_tune_loss_variants(params, case, kw):
    tr = trainer(params, **kw)
    data, mask, pred, sigma = g("nll/inputs", "data", "mask", "pred", "sigma")
    n = len(mask)
    y_true = dev(np.concatenate([data, mask[:, None]], -1).reshape(n, 1, 1, 1, 12))
    y_pred = dev(np.concatenate([pred, sigma], -1).reshape(n, 1, 1, 1, 22))
    close(tr.fine_tune_loss_fn(y_true, y_pred), g(f"nll/{case}", "mean"), rtol=2e-5)
    close(tr.fine_tune_loss_fn(y_true, y_pred, return_mean=False).reshape(n), g(f"nll/{case}", "per_voxel"), rtol=3e-5,
          atol=1e-4)


def test_fine_tune_loss_tiled_batch(params):
    data, mask, pred, sigma = g("nll/inputs", "data", "mask", "pred", "sigma")
    preds, mean, rows = g("nll/three_samples", "preds", "mean", "per_row")
    n = len(mask)
    tr = trainer(params, no_samples=3)
    y_true = dev(np.concatenate([data, mask[:, None]], -1).reshape(n, 1, 1, 1, 12))
    y_pred = dev(np.concatenate([np.concatenate([preds[:, s], sigma], -1) for s in range(3)], 0).reshape(3 * n, 1, 1, 1, 22))
    close(tr.fine_tune_loss_fn(y_true, y_pred), mean, rtol=2e-5)
    close(tr.fine_tune_loss_fn(y_true, y_pred, return_mean=False).reshape(3, n), rows, rtol=3e-5, atol=1e-4)


def test_kl_terms_and_smoothness(params):
    tr, td = trainer(params), trainer(params, use_mvg=False)
    q, prior, mask, z = g("kl/sampled", "q", "prior", "mask", "z")
    n = len(q)
    kl = tr.context.kl_fwd(dev(q), dev(prior), K=70, zk=dev(z)) * dev((mask > 0).astype(np.float32))
    close(kl, g("kl/sampled", "per_voxel"), rtol=1e-4, atol=1e-4)
    close(kl.double().sum() / float(mask.sum()), g("kl/sampled", "mean"), rtol=1e-4)
    true6 = dev(np.concatenate([prior, mask[:, None]], -1).reshape(n, 1, 1, 1, 6))
    q5 = dev(q.reshape(n, 1, 1, 1, 5))
    close(tr.mvg_kl(true6, q5).reshape(n), g("kl/sampled", "closed_form"), rtol=1e-4, atol=1e-4)
    # the sampled estimator on the library's own Philox draws: same expectation (model.py:592-610)
    own = tr.kl_loss(true6, q5, no_samples=70, seed=5)
    assert abs(float(own) / float(g("kl/sampled", "mean")) - 1) < 0.05
    true5 = dev(np.concatenate([prior[:, :4], mask[:, None]], -1).reshape(n, 1, 1, 1, 5))
    close(td.kl_loss(true5, q5[..., :4]), g("kl/diag", "mean"), rtol=1e-4)
    close(td.kl_loss(true5, q5[..., :4], return_mean=False).reshape(n), g("kl/diag", "per_voxel"), rtol=1e-4, atol=1e-5)
    qc, mc, loss, diag_loss = g("smoothness", "q", "mask", "loss", "diag_loss")
    truec = dev(np.concatenate([qc, mc[..., None]], -1))
    close(tr.smoothness_loss(truec, dev(qc)), loss, rtol=2e-5)
    close(td.smoothness_loss(truec[..., [0, 1, 2, 3, 5]], dev(qc[..., :4])), diag_loss, rtol=2e-5)


# -- create_encoder / build_fine_tuner ----------------------------------------------------------------
@pytest.mark.parametrize("case", ["encoder_relu", "encoder_shared_gate"])
def test_two_stream_encoder(params, case):
    tr, model = encoder_of(params, case)
    x, o1, o2, sg = g(f"{case}/voxels", "x", "out1", "out2", "sigma")
    g1, g2, gs = model(dev(x.reshape(-1, 1, 1, 1, 11)))
    close(g1.reshape(o1.shape), o1, rtol=2e-5, atol=2e-5)
    close(g2.reshape(o2.shape), o2, rtol=2e-5, atol=2e-5)
    close(gs.reshape(sg.shape), sg, rtol=5e-5)
    xc, c1, c2, cs = g(f"{case}/crops", "x", "out1", "out2", "sigma")
    s1, s2, ssg = model(dev(xc))
    close(s1, c1, rtol=2e-5, atol=2e-5)
    close(s2, c2, rtol=2e-5, atol=2e-5)
    close(ssg, cs, rtol=5e-5)


@pytest.mark.parametrize("case,act", [("encoder_layer_norm_relu", "relu"), ("encoder_layer_norm_gelu", "gelu")])
def test_encoder_with_layer_norm_and_dropout(params, case, act):
    """EncoderTrainer(use_layer_norm=True, dropout_rate=0.2) through the reference-shaped API against what the
    reference text computed (model.py:131-140): GroupNormalization over a voxel's channels on (N,1,1,1,T) batches and over
    a whole crop on crops, Dropout the identity outside training."""
    from oracle.oracle import WEIGHT_NAMES
    U, L = int(G[f"{case}/weights/U"]), int(G[f"{case}/weights/L"])
    tr = trainer(params, no_units=U, no_intermediate_layers=L, channelwise_gating=True, use_layer_norm=True,
                 dropout_rate=0.2, activation_type=act)
    model, _ = tr.create_encoder(gate_offset=float(G[f"{case}/weights/gate_offset"]), resid_init_std=0.05, no_ip_images=11)
    model.set_weights(dict({n: G[f"{case}/weights/{n}"] for n in WEIGHT_NAMES}, ln=G[f"{case}/weights/ln"]))
    x, o1, o2, sg = g(f"{case}/voxels", "x", "out1", "out2", "sigma")
    g1, g2, gs = model(dev(x.reshape(-1, 1, 1, 1, 11)))
    close(g1.reshape(o1.shape), o1, rtol=2e-5, atol=2e-5)
    close(g2.reshape(o2.shape), o2, rtol=3e-5, atol=3e-5)
    close(gs.reshape(sg.shape), sg, rtol=1e-4)
    xc, c1, c2, cs = g(f"{case}/crops", "x", "out1", "out2", "sigma")
    s1, s2, ssg = model(dev(xc))
    close(s1, c1, rtol=2e-5, atol=2e-5)
    close(s2, c2, rtol=3e-5, atol=3e-5)
    close(ssg, cs, rtol=1e-4)


def test_fine_tuner_elbo_from_the_reference_text(params):
    """build_fine_tuner on a crop batch with S = 2, then fine_tune_loss_fn + kl_loss (train.py:315-320): encoder with
    3x3x1 context -> draws -> forward model -> NLL + KL, on the reference's own normals."""
    from qbold_vi_amd.signals import SignalGenerationLayer
    data, mask, prior, zs, zk = g("fine_tuner", "data", "mask", "prior", "zs", "zk")
    S = int(g("fine_tuner", "S"))
    tr, model = encoder_of(params, "fine_tuner", no_samples=S)
    B, X, Y, Z, T = data.shape
    n = B * X * Y * Z
    o1, q, sigma = model(dev(data))
    close(o1, prior, rtol=2e-5, atol=2e-5)
    pred, imgs = g("fine_tuner", "predictions", "predicted_images")
    close(torch.cat([q] * S), pred, rtol=2e-5, atol=2e-5)
    close(torch.cat([sigma] * S), imgs[..., T:], rtol=5e-5)
    # the reference-shaped composition: the product's FineTuner pieces on the reference's draws
    full = tr.build_fine_tuner(model, SignalGenerationLayer(dict(params, simulate_noise="False"), True, True))
    sampled = full._rpl((torch.cat([q] * S), None), z=dev(zs.reshape(-1, 2)))
    signal = full.signal_generation_layer(sampled)
    close(signal, imgs[..., :T], rtol=3e-5)
    y_true = dev(np.concatenate([data, mask[..., None]], -1))
    y_pred = torch.cat([signal, torch.cat([sigma] * S)], -1)
    close(tr.fine_tune_loss_fn(y_true, y_pred), g("fine_tuner", "nll"), rtol=5e-5)
    close(tr.fine_tune_loss_fn(y_true, y_pred, return_mean=False).reshape(S * B, X, Y, Z), g("fine_tuner", "nll_rows"),
          rtol=5e-5, atol=1e-3)
    # the fused evaluation (one ELBO kernel on the spatial encoder's heads) on the same draws
    K = zk.shape[0]
    zs_v = zs.reshape(S, n, 2).transpose(1, 0, 2)
    zk_v = zk.reshape(K, S, n, 2).transpose(2, 1, 0, 3).reshape(n, S * K, 2)
    sums, nk = tr.context.elbo_fwd(dev(data.reshape(n, T)), dev(mask.reshape(n)), q.reshape(n, 5).contiguous(),
                                   dev(prior.reshape(n, 5)), sigma.reshape(n, T).contiguous(), S, S * K,
                                   zs=dev(zs_v), zk=dev(zk_v))
    nll, kl = float(sums[0] / sums[2]), float(sums[1] / sums[2])
    close(nll, g("fine_tuner", "nll"), rtol=1e-4)
    close(kl, g("fine_tuner", "kl"), rtol=1e-4)
    close(nll + kl, g("fine_tuner", "neg_elbo"), rtol=1e-4)                 # the north-star's ELBO tolerance
    nll_rows, kl_rows = g("fine_tuner", "nll_rows", "kl_rows")
    m = mask.reshape(n)
    close(nk[:, 0].cpu().numpy() * m, nll_rows.reshape(S, n).mean(0), rtol=5e-5, atol=1e-3)
    close(nk[:, 1].cpu().numpy() * (m > 0), kl_rows.reshape(S, n).mean(0), rtol=1e-4, atol=1e-4)
    true_c = dev(np.concatenate([prior, mask[..., None]], -1))
    close(tr.smoothness_loss(true_c, torch.cat([q] * S)), g("fine_tuner", "smoothness"), rtol=2e-5)


def test_population_prior_kl_diag(params):
    """kl_loss with use_population_prior and the diagonal family (model.py:687-716)."""
    q, prior, mask = g("kl/sampled", "q", "prior", "mask")
    pop = g("kl/population_diag", "pop_prior")
    n = len(q)
    tp = trainer(params, use_mvg=False, use_population_prior=True, mog_components=1)
    true5 = dev(np.concatenate([prior[:, :4], mask[:, None]], -1).reshape(n, 1, 1, 1, 5))
    pred8 = dev(np.concatenate([q[:, :4], np.broadcast_to(pop, (n, 4))], -1).reshape(n, 1, 1, 1, 8))
    close(tp.kl_loss(true5, pred8), g("kl/population_diag", "mean"), rtol=1e-4)
    close(tp.kl_loss(true5, pred8, return_mean=False).reshape(n), g("kl/population_diag", "per_voxel"), rtol=1e-4, atol=1e-5)
    with pytest.raises(NotImplementedError, match="shape error in the reference"):
        trainer(params, use_mvg=True, use_population_prior=True)


def test_homoscedastic_fine_tuner(params):
    """heteroscedastic_noise=False (model.py:277-281, 535-537)."""
    from qbold_vi_amd.signals import SignalGenerationLayer
    data, zs, imgs = g("fine_tuner_homoscedastic", "data", "zs", "predicted_images")
    s0 = float(g("fine_tuner_homoscedastic", "initial_im_sigma"))
    tr, model = encoder_of(params, "fine_tuner_homoscedastic", heteroscedastic_noise=False, initial_im_sigma=s0)
    full = tr.build_fine_tuner(model, SignalGenerationLayer(dict(params, simulate_noise="False"), True, True))
    n = len(data)
    x5 = dev(data.reshape(n, 1, 1, 1, 11))
    m5 = torch.ones(n, 1, 1, 1, 1, device="cuda")
    out = full([x5, m5])                                    # the library's own draws: shapes and the sigma channel
    assert tuple(out["predicted_images"].shape) == (n, 1, 1, 1, 12) and tuple(out["predictions"].shape) == (n, 1, 1, 1, 5)
    close(out["predicted_images"][..., 11].reshape(n), np.full(n, s0), rtol=1e-6)
    # the reference's draws
    q = model(x5)[1]
    signal = full.signal_generation_layer(full._rpl((q, None), z=dev(zs)))
    close(signal.reshape(n, 11), imgs[:, :11], rtol=3e-5)
    y_pred = torch.cat([signal, torch.full((n, 1, 1, 1, 1), s0, device="cuda")], -1)
    close(tr.fine_tune_loss_fn(torch.cat([x5, m5], -1), y_pred), g("fine_tuner_homoscedastic", "nll"), rtol=2e-5)
    # the fused evaluation uses the scalar, not the encoder's sigma head
    prior = model(x5)[0].reshape(n, 5)
    e = full.elbo(x5, m5, prior, no_samples=64, kl_samples=8, seed=3, kl_tiled=False)
    sg = torch.full((n, 11), s0, device="cuda")
    sums, _ = tr.context.elbo_fwd(dev(data), None, q.reshape(n, 5).contiguous(), prior, sg, 64, 8, seed=3)
    assert torch.equal(e["sums"], sums)


def test_gelu_encoder(params):
    """activation_type='gelu' (model.py:60, 115-120, 151, 155): forward on voxel batches and crops."""
    tr, model = encoder_of(params, "encoder_gelu", activation_type="gelu")
    x, o1, o2, sg = g("encoder_gelu/voxels", "x", "out1", "out2", "sigma")
    g1, g2, gs = model(dev(x.reshape(-1, 1, 1, 1, 11)))
    close(g1.reshape(o1.shape), o1, rtol=2e-5, atol=2e-5)
    close(g2.reshape(o2.shape), o2, rtol=2e-5, atol=2e-5)
    close(gs.reshape(sg.shape), sg, rtol=5e-5)
    xc, c1, c2, cs = g("encoder_gelu/crops", "x", "out1", "out2", "sigma")
    s1, s2, ssg = model(dev(xc))
    close(s1, c1, rtol=2e-5, atol=2e-5)
    close(s2, c2, rtol=2e-5, atol=2e-5)
    close(ssg, cs, rtol=5e-5)
    # the ELBO evaluation composes the same forward with the sampling kernel
    from qbold_vi_amd.signals import SignalGenerationLayer
    full = tr.build_fine_tuner(model, SignalGenerationLayer(dict(params, simulate_noise="False"), True, True))
    n = len(x)
    e = full.elbo(dev(x), None, g1.reshape(n, 5), kl_samples=16, seed=2)
    sums, _ = tr.context.elbo_fwd(dev(x), None, g2.reshape(n, 5).contiguous(), g1.reshape(n, 5).contiguous(),
                                  gs.reshape(n, 11).contiguous(), 1, 16, seed=2)
    assert torch.equal(e["sums"], sums)
    # (its training gradients: tests/test_gpu_branches.py::test_gelu_training_gradients_*)


def test_mixture_of_gaussians_kl(params):
    """kl_loss with mog_components = 3 (model.py:666-685) on the reference's own draws, and through the host mirror."""
    q, prior, mask = g("kl/sampled", "q", "prior", "mask")
    comps, z = g("kl/mog", "components", "z")
    n = len(q)
    tm = trainer(params, use_mvg=False, use_population_prior=True, mog_components=3)
    kl = tm.context.kl_mog(dev(q), dev(comps), z=dev(z)) * dev((mask > 0).astype(np.float32))
    close(kl, g("kl/mog", "per_voxel"), rtol=1e-4, atol=1e-5)
    close(kl.double().sum() / float(mask.sum()), g("kl/mog", "mean"), rtol=1e-4)
    # the reference-shaped call draws its own normals: same expectation
    true5 = dev(np.concatenate([prior[:, :4], mask[:, None]], -1).reshape(n, 1, 1, 1, 5))
    pred = dev(np.concatenate([q[:, :4], np.broadcast_to(comps.reshape(-1), (n, 12))], -1).reshape(n, 1, 1, 1, 16))
    vals = [float(tm.kl_loss(true5, pred, seed=s)) for s in range(1, 9)]
    assert abs(np.mean(vals) / float(g("kl/mog", "mean")) - 1) < 0.25 and np.std(vals) > 0
    assert tuple(tm.kl_loss(true5, pred, return_mean=False).shape) == (n, 1, 1, 1, 1)
    # the fine-tuner appends the 4 M prior values to 'predictions' and evaluates the ELBO with this KL
    from qbold_vi_amd.signals import SignalGenerationLayer
    model, _ = tm.create_encoder(gate_offset=-3.0, resid_init_std=0.05, no_ip_images=11)
    full = tm.build_fine_tuner(model, SignalGenerationLayer(dict(params, simulate_noise="False"), True, True))
    x = g("nll/inputs", "data")[:64]
    out = full([dev(x.reshape(64, 1, 1, 1, 11)), None])
    assert tuple(out["predictions"].shape) == (64, 1, 1, 1, 16)
    e = full.elbo(dev(x), None, dev(prior[:64]), seed=4)
    assert np.isfinite(float(e["kl"])) and np.isfinite(float(e["nll"]))

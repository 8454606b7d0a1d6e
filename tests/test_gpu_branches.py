"""The branches configurations/optimal.yaml leaves off, trained: the homoscedastic sigma
(heteroscedastic_noise=False, model.py:277-281) and what the population prior can and cannot do."""
import math
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(params, **kw):
    from qbold_vi_amd.model import EncoderTrainer
    from qbold_vi_amd.signals import SignalGenerationLayer
    from qbold_vi_amd.training import prepare_voxel_dataset, synthetic_voxel_dataset
    base = dict(no_intermediate_layers=1, no_units=24, activation_type="relu", student_t_df=200, initial_im_sigma=0.08,
                multi_image_normalisation=False, channelwise_gating=True, use_mvg=True, use_population_prior=False,
                no_samples=2, predict_log_data=False)
    base.update(kw)
    tr = EncoderTrainer(params, **base)
    model, _ = tr.create_encoder(gate_offset=-3.0, resid_init_std=0.05, no_ip_images=11)
    cfg = dict(full_model=True, use_blood=True, uniform_prop=0.0)
    x, mask, _ = synthetic_voxel_dataset(params, cfg, 8192, tr.context.device, seed=5)
    mask = (torch.rand(8192, device=x.device, generator=torch.Generator(device=x.device).manual_seed(1)) > 0.2).float()
    full = tr.build_fine_tuner(model, SignalGenerationLayer(dict(params, simulate_noise="False"), True, True))
    return tr, model, full, prepare_voxel_dataset(x, mask, model)


def test_homoscedastic_sigma_gradient_and_training(params):
    from qbold_vi_amd import training
    tr, model, full, (x, mask, prior) = _setup(params, heteroscedastic_noise=False)
    assert full.log_sigma == pytest.approx(math.log(0.08))
    # d(-ELBO) / d log sigma: the sum of the per-tau head gradients over sum(mask) against a central difference of the
    # evaluation at a fixed Philox seed
    q = model(x)[1]
    ls = torch.full((x.shape[0], 11), full.log_sigma, device=x.device)
    sums, gq, gls = training._elbo_bwd(tr, x, mask, q.contiguous(), prior, ls, 2, 10, 77, 0)
    g = float(gls.sum(dtype=torch.float64) / sums[2])
    h, vals = 1e-3, []
    for d in (+h, -h):
        full.log_sigma = math.log(0.08) + d
        vals.append(float(full.elbo(x, mask, prior, no_samples=2, kl_samples=10, seed=77)["elbo"]))
    full.log_sigma = math.log(0.08)
    fd = (vals[0] - vals[1]) / (2 * h)
    assert abs(g - fd) < 2e-3 * max(1.0, abs(fd)), (g, fd)
    # a short fine-tuning run: the scalar moves (towards the residual level of this barely trained encoder), the encoder's
    # own sigma head (outside the trained graph: gradient None) keeps its initial values bit for bit, the objective falls
    w0 = model.get_weights()
    cfg = dict(adamw_decay=2e-4, ft_lr=5e-3, no_ft_epochs=3, smoothness_weight=0.0)
    hist = training.MetricsLog(echo=False)
    training.train_full_model(cfg, tr, full, (x[:1024], mask[:1024], prior[:1024]), (x, mask, prior), log=hist,
                              steps_per_epoch=20, batch_voxels=4096, kl_samples=10)
    w1 = model.get_weights()
    np.testing.assert_array_equal(w0["Ws"], w1["Ws"])
    np.testing.assert_array_equal(w0["bs"], w1["bs"])
    assert np.abs(w0["Wf"] - w1["Wf"]).max() > 0
    assert abs(full.log_sigma - math.log(0.08)) > 0.02 and math.isfinite(full.log_sigma)
    h = hist.history
    assert h[-1]["loss"] < h[0]["loss"] and all(math.isfinite(v["val_elbo"]) for v in h)


def test_population_prior_evaluates_but_cannot_fine_tune(params):
    """The reference's fine-tuning loss always contains smoothness_loss (train.py:318-320), which fails on the
    population prior's 8-channel predictions (model.py:729-739): evaluation is built, training raises."""
    from qbold_vi_amd import training
    tr, model, full, (x, mask, prior) = _setup(params, use_mvg=False, use_population_prior=True, mog_components=1,
                                               no_samples=1)
    out = full([x[:256].reshape(256, 1, 1, 1, 11), None])
    assert tuple(out["predictions"].shape) == (256, 1, 1, 1, 8)
    np.testing.assert_allclose(out["predictions"][0, 0, 0, 0, 4:].cpu().numpy(), [-0.97, 0.4, -1.14, 0.6], rtol=1e-6)
    e = full.elbo(x, mask, prior, kl_samples=70, seed=1)
    # KL term = (sum of closed-form KLs to the ONE population prior + inverse-gamma cost x batch) / sum(mask)
    pp = torch.as_tensor(full.pop_prior, device=x.device).expand(x.shape[0], 4)
    pp5 = torch.cat([pp, torch.zeros_like(pp[:, :1])], -1).contiguous()
    q5 = torch.cat([e["q"], torch.zeros_like(e["q"][:, :1])], -1).contiguous()
    ksums, _ = tr.context.kl_diag(q5, pp5, mask)
    want = (float(ksums[1]) + tr.population_prior_cost(full.pop_prior, x.shape[0])) / float(mask.sum())
    assert abs(float(e["kl"]) - want) < 1e-9 * abs(want)
    cfg = dict(adamw_decay=2e-4, ft_lr=5e-3, no_ft_epochs=1, smoothness_weight=0.0)
    with pytest.raises(NotImplementedError, match="smoothness_loss"):
        training.train_full_model(cfg, tr, full, (x, mask, prior), (x, mask, prior), steps_per_epoch=2)


@pytest.fixture()
def gelu_oracles(params):
    """The oracle's activation switch is process-global in the C library: set for the test, restored after."""
    from oracle.oracle import Oracle
    o32 = Oracle("f32", params)
    o64 = Oracle("f64", params, node0_zero=True)
    o32.set_activation("gelu")
    o64.set_activation("gelu")
    yield o32, o64
    o32.set_activation("relu")
    o64.set_activation("relu")
    o64.lib.qbo_set_node0_zero(0)


def _gelu_weights(ctx, U, L, cw, taps):
    from oracle.oracle import init_weights
    from qbold_vi_amd.ops import EncoderWeights
    w = init_weights(T=11, U=U, L=L, channelwise_gating=cw, seed=5, taps=taps, resid_init_std=0.08)
    rng = np.random.default_rng(5)
    for k in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
    w["gate_offset"] = -1.0
    ew = EncoderWeights(ctx, 11, U, L, cw, -1.0, spatial_taps=taps, activation="gelu").set_from_arrays(w)
    return w, ew


@pytest.mark.parametrize("U,L,cw", [(24, 2, True), (20, 1, False), (80, 1, True)])
def test_gelu_training_gradients_voxel_batches(params, gelu_oracles, U, L, cw):
    """activation_type='gelu' (model.py:60, 115-120, 151, 155) under training: the backward recomputes every
    pre-activation and multiplies by gelu'(z).  Both streams, against central differences of the float64 oracle along
    random weight directions (gelu is smooth: no kinks to cross)."""
    from oracle.oracle import WEIGHT_NAMES, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
    from test_gpu_grad import _perturbed, kl_stopgrad
    o32, o64 = gelu_oracles
    ctx = Context(params, True, True)
    ctx.set_grad_node0(False)
    w, ew = _gelu_weights(ctx, U, L, cw, 1)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    rng = np.random.default_rng(3)

    def directional(grad, loss, tol, trials=3, freeze=()):
        for trial in range(trials):
            direction = {k: rng.standard_normal(w[k].shape) for k in WEIGHT_NAMES}
            for k in freeze:
                direction[k] *= 0
            dflat = EncoderWeights(ctx, 11, U, L, cw, -1.0).set_from_arrays(
                {k: direction[k].astype(np.float32) for k in WEIGHT_NAMES}).flat.cpu().numpy().astype(np.float64)
            eps = 1e-4
            fd = (loss(_perturbed(w, direction, eps)) - loss(_perturbed(w, direction, -eps))) / (2 * eps)
            got = float(grad @ dflat)
            assert abs(got - fd) < tol * (abs(fd) + 0.05), (trial, got, fd)

    # stream 1 / pre-training loss
    n = 384
    x, y = synth_inputs(n, seed=4, oracle=o32)
    y3 = np.concatenate([y, y[:, :1]], -1).astype(np.float32)
    st = TrainState(ctx, ew)
    q1, _ = st.forward(dev(x), 1)
    close = np.abs(q1.cpu().numpy() - o32.encoder_fwd(w, x)[0]).max()
    assert close < 2e-5, close
    lv, gq = st.synth_loss_bwd(dev(y3), q1)
    grad = st.backward(1, gq).cpu().numpy().astype(np.float64)
    directional(grad, lambda ww: o64.synthetic_data_loss(y3, o64.encoder_fwd(ww, x)[0]), 5e-3,
                freeze=("Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Ws", "bs"))
    # stream 2 / negative ELBO
    S, K, seed = 2, 6, 21
    mask = (rng.uniform(size=n) > 0.2).astype(np.float32)
    prior = o32.encoder_fwd(w, x)[0]
    q2, ls = st.forward(dev(x), 2)
    sums, gq, gls, _ = ctx.elbo_bwd(dev(x), dev(mask), q2, dev(prior), ls, S, K, seed=seed)
    grad = st.backward(2, gq, gls, sums).cpu().numpy().astype(np.float64)
    zs, zk = o32.philox_normals(seed, 0, 0, n, S), o32.philox_normals(seed, 1, 0, n, K)
    q_fixed = o64.encoder_fwd(w, x)[1]

    def loss2(ww):
        _, qq, sg = o64.encoder_fwd(ww, x)
        e = o64.elbo(x, mask, qq, prior, sg, zs, zk)
        kl = kl_stopgrad(o64, qq, q_fixed, prior, zk)
        return ((e["nll_v"] * mask).sum() + np.where(mask > 0, kl, 0).sum()) / mask.sum()
    directional(grad, loss2, 1e-2)


def test_gelu_training_gradients_crops(params, gelu_oracles):
    """The same on image crops: 3x3x1 convolutions, every tap's gradient."""
    from oracle.oracle import WEIGHT_NAMES, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
    from test_gpu_grad import _perturbed, kl_stopgrad
    o32, o64 = gelu_oracles
    ctx = Context(params, True, True)
    ctx.set_grad_node0(False)
    U, L, cw = 20, 2, True
    w, ew = _gelu_weights(ctx, U, L, cw, 9)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    B, X, Y, Z = 2, 5, 4, 2
    n, S, K, seed = B * X * Y * Z, 2, 4, 9
    x = synth_inputs(n, seed=4, oracle=o32)[0].reshape(B, X, Y, Z, 11)
    rng = np.random.default_rng(5)
    mask = (rng.uniform(size=(B, X, Y, Z)) > 0.2).astype(np.float32)
    xf, mf = x.reshape(n, 11), mask.reshape(n)
    wc = dict(w, Wr1=w["Wr1"][:, 1, 1], Wr2=w["Wr2"][:, 1, 1], meta=dict(w["meta"], taps=1))
    prior = o32.encoder_fwd(wc, xf)[0]
    st = TrainState(ctx, ew)
    q, ls = st.forward_spatial(dev(x))
    want_q, want_sg = o32.encoder_fwd_spatial(w, x)
    assert np.abs(q.cpu().numpy() - want_q.reshape(n, 5)).max() < 2e-5
    sums, gq, gls, _ = ctx.elbo_bwd(dev(xf), dev(mf), q, dev(prior), ls, S, K, seed=seed)
    grad = st.backward_spatial(gq, gls, sums).cpu().numpy().astype(np.float64)
    zs, zk = o32.philox_normals(seed, 0, 0, n, S), o32.philox_normals(seed, 1, 0, n, K)
    q_fixed = o64.encoder_fwd_spatial(w, x)[0].reshape(n, 5)

    def loss(ww):
        o2, sg = o64.encoder_fwd_spatial(ww, x)
        qq = o2.reshape(n, 5)
        e = o64.elbo(xf, mf, qq, prior, sg.reshape(n, 11), zs, zk)
        kl = kl_stopgrad(o64, qq, q_fixed, prior, zk)
        return ((e["nll_v"] * mf).sum() + np.where(mf > 0, kl, 0).sum()) / mf.sum()

    for trial in range(3):
        direction = {k: rng.standard_normal(w[k].shape) for k in WEIGHT_NAMES}
        if trial == 1:   # only the off-centre taps
            for k in WEIGHT_NAMES:
                direction[k] *= 0 if k not in ("Wr1", "Wr2") else 1
            direction["Wr1"][:, 1, 1] = 0
            direction["Wr2"][:, 1, 1] = 0
        dflat = EncoderWeights(ctx, 11, U, L, cw, -1.0, spatial_taps=9).set_from_arrays(
            {k: direction[k].astype(np.float32) for k in WEIGHT_NAMES}).flat.cpu().numpy().astype(np.float64)
        eps = 1e-4
        fd = (loss(_perturbed(w, direction, eps)) - loss(_perturbed(w, direction, -eps))) / (2 * eps)
        got = float(grad @ dflat)
        assert abs(got - fd) < 1e-2 * (abs(fd) + 0.05), (trial, got, fd)


def test_gelu_two_phase_training_runs(tmp_path, monkeypatch):
    """`activation: gelu` through the reference-shaped entry point: both phases run and learn."""
    from qbold_vi_amd import training
    from qbold_vi_amd.utils import load_arguments
    monkeypatch.chdir(ROOT)
    cfg = load_arguments(["train.py", os.path.join(ROOT, "configurations", "optimal.yaml")], entry="train")
    cfg.update(no_units=24, no_intermediate_layers=1, no_pt_epochs=30, no_ft_epochs=2, save_directory=str(tmp_path),
               synthetic_voxels=20000, mc_samples=2, activation="gelu")
    model, trainer, hist = training.train_model(cfg, pt_sample_size=200, max_ft_steps=40)
    pt = [h for h in hist if "val_oef_metric" in h]
    ft = [h for h in hist if "val_elbo" in h]
    assert trainer._activation_type == "gelu" and len(pt) == 30 and ft
    assert pt[-1]["loss"] < pt[0]["loss"] - 5.0
    assert all(math.isfinite(h["val_elbo"]) for h in ft)

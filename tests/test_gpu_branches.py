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

"""Training orchestration -- the host mirror of train.py:188-451 and qbold_train_model.py /
qbold_build_model.py of the reference, for voxel batches.

Two phases, as the reference: (1) supervised pre-training of encoder stream 1 on synthetic voxels
(create_and_train_on_synthetic_data, train.py:379-427), (2) unsupervised ELBO fine-tuning of
stream 2 with the stream-1 predictions as per-voxel prior (train_full_model, train.py:285-376).
Weights are checkpointed as <save_directory>/pt_model.npz and final_model.npz with the reference's
three-state phase skipping (qbold_build_model.py:11-56).  Every number is produced by
libqbold_hip.so; with N > 1 ranks (torchrun) voxels are sharded and gradients / loss sums are
all-reduced over RCCL (qbold_vi_amd.distributed).

Fine-tuning data: real `.npy` volumes go through the reference's random-crop pipeline
(prepare_dataset, train.py:17-72: 38 crops of crop_size x crop_size x Z per step, stream 2 with its
3x3x1 convolutions, TV smoothness term) on the layer-wise spatial kernels; `--synthetic_voxels N`
fine-tunes on N i.i.d. synthetic voxels through the fused voxel kernels instead.  wandb is replaced
by JSONL metrics with the reference's key names.
"""
import json
import math
import os
import time
from enum import Enum

import numpy as np
import torch

from . import distributed as qd
from .model import EncoderTrainer, R2P_LOSS_SAMPLES
from .ops import TrainState
from .signals import SignalGenerationLayer, create_synthetic_dataset


class WeightStatus(Enum):  # qbold_build_model.py:11-14
    NOT_TRAINED = 0
    PRE_TRAINED = 1
    FULL_TRAINED = 2


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def lr_schedule(initial, step, steps_per_epoch=100):
    """LRSchedule.__call__ (train.py:287-306): linear from `initial` to initial/100 over 40 epochs of
    100 steps; `step` is the optimiser's iteration count before the update."""
    final = initial / 1e2
    rate = (final - initial) / (40.0 * steps_per_epoch)
    return initial + rate * step if step > 0 else initial


def get_params(path="config"):
    """The INI `config` [DEFAULT] section, read from the CWD as strings (train.py:189-191)."""
    import configparser
    cfg = configparser.ConfigParser()
    if not cfg.read(path):
        raise FileNotFoundError(f"INI file {path!r} not found in {os.getcwd()}")
    return dict(cfg["DEFAULT"])


def _existing_weights(save_dir, stem):
    """<save_dir>/<stem>.npz (native).  The reference's Keras <stem>.h5 (qbold_build_model.py:29-33) is read
    through keras_h5.py only on request -- QBOLD_KERAS_H5=1 -- because that mapping has never met a real Keras
    file (no h5py, no TensorFlow, no shipped weights here): experimental until it has."""
    npz, h5 = os.path.join(save_dir, stem + ".npz"), os.path.join(save_dir, stem + ".h5")
    if os.environ.get("QBOLD_KERAS_H5") == "1" and os.path.isfile(h5) and not os.path.isfile(npz):
        return h5
    return npz


def create_encoder_model(config_dict, params, device=None):
    """train.create_encoder_model (train.py:430-451) = ModelBuilder.create_encoder_model
    (qbold_build_model.py:59-82).  Returns (model, inner_model, trainer)."""
    nl = max(1, int(_get(config_dict, "no_intermediate_layers")))
    nu = max(1, int(_get(config_dict, "no_units")))
    trainer = EncoderTrainer(system_params=params,
                             no_units=nu,
                             use_layer_norm=_get(config_dict, "use_layer_norm"),
                             dropout_rate=_get(config_dict, "dropout_rate"),
                             no_intermediate_layers=nl,
                             student_t_df=_get(config_dict, "student_t_df"),
                             initial_im_sigma=_get(config_dict, "im_loss_sigma"),
                             activation_type=_get(config_dict, "activation"),
                             multi_image_normalisation=_get(config_dict, "multi_image_normalisation"),
                             channelwise_gating=_get(config_dict, "channelwise_gating"),
                             infer_inv_gamma=_get(config_dict, "infer_inv_gamma"),
                             use_population_prior=_get(config_dict, "use_population_prior"),
                             use_mvg=_get(config_dict, "use_mvg"),
                             heteroscedastic_noise=_get(config_dict, "heteroscedastic_noise", True),
                             predict_log_data=_get(config_dict, "predict_log_data"),
                             no_samples=max(1, int(_get(config_dict, "mc_samples", 1) or 1)),
                             full_model=_get(config_dict, "full_model", True),
                             use_blood=_get(config_dict, "use_blood", True),
                             device=device)
    taus = np.arange(float(params['tau_start']), float(params['tau_end']), float(params['tau_step']))
    model, inner_model = trainer.create_encoder(gate_offset=_get(config_dict, "gate_offset"),
                                                resid_init_std=_get(config_dict, "resid_init_std"),
                                                no_ip_images=len(taus))
    return model, inner_model, trainer


class MetricsLog:
    """wandb.log / Keras progress replacement: one JSON object per line, reference key names."""

    def __init__(self, path=None, echo=True, rank=0):
        self.path, self.echo, self.rank = path, echo, rank
        self.history = []

    def log(self, metrics):
        m = {k: (float(v) if not isinstance(v, (str, int)) else v) for k, v in metrics.items()}
        self.history.append(m)
        if self.rank != 0:
            return
        line = json.dumps(m)
        if self.echo:
            print(line, flush=True)
        if self.path:
            with open(self.path, "a") as fh:
                fh.write(line + "\n")


def _check_finite(value, what):
    if not math.isfinite(value):  # keras.callbacks.TerminateOnNaN (train.py:375,423)
        raise FloatingPointError(f"{what} is not finite: training terminated")


# ----------------------------------------------------------------------------------------------
# phase 1: pre-training on synthetic data (train.py:379-427)
# ----------------------------------------------------------------------------------------------
def digamma(x):
    """psi(x) for x > 0 in float64: the recurrence psi(x) = psi(x + 1) - 1 / x up to x >= 10, then the asymptotic
    series ln x - 1/(2x) - 1/(12x^2) + 1/(120x^4) - 1/(252x^6) + 1/(240x^8) - 1/(132x^10) (next term 0.021 / x^12 < 3e-14)."""
    x = float(x)
    if not x > 0.0:
        raise ValueError("digamma: the inverse-gamma hyper-parameters are positive")
    acc = 0.0
    while x < 10.0:
        acc -= 1.0 / x
        x += 1.0
    f = 1.0 / (x * x)
    return acc + math.log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f / 132))))


class HyperPriorState:
    """The four exp-activated scalars of infer_inv_gamma (model.py:201-205) under the pre-training optimiser
    (tfa AdamW, Keras Adam moments, eps 1e-7; train.py:382-385).  With theta = log of (a_o, b_o, a_d, b_d) and the
    batch means L = mean log v, R = mean 1 / v of a dimension, the loss term is
    lgamma(a) - a log b + (a + 1) L + b R, so d / d a = digamma(a) - log b + L, d / d b = - a / b + R, and the
    chain rule through exp multiplies by a (resp. b)."""

    def __init__(self, model):
        self.model = model
        self.m = np.zeros(4)
        self.v = np.zeros(4)
        self.t = 0

    def gradient(self, stats, n):
        a_o, b_o, a_d, b_d = self.model.hyper_params()
        Lo, Ro, Ld, Rd = (float(s) / n for s in stats)
        return np.array([a_o * (digamma(a_o) - math.log(b_o) + Lo), b_o * (-a_o / b_o + Ro),
                         a_d * (digamma(a_d) - math.log(b_d) + Ld), b_d * (-a_d / b_d + Rd)])

    def step(self, stats, n, lr, wd, beta1=0.9, beta2=0.999, eps=1e-7):
        g = self.gradient(stats, n)
        self.t += 1
        self.m = beta1 * self.m + (1 - beta1) * g
        self.v = beta2 * self.v + (1 - beta2) * g * g
        lr_t = lr * math.sqrt(1 - beta2 ** self.t) / (1 - beta1 ** self.t)
        th = self.model.hyper_raw
        th = th - wd * th
        self.model.hyper_raw = th - lr_t * self.m / (np.sqrt(self.v) + eps)


class ScalarAdamW:
    """One host-side scalar under the fine-tuning optimiser (tfa AdamW: decoupled decay, Keras Adam moments and
    bias correction, eps 1e-7) -- the exp-activated sigma variable of heteroscedastic_noise=False
    (model.py:277-281), which belongs to the fine-tuner, not to the encoder's weight blob."""

    def __init__(self, value):
        self.value, self.m, self.v, self.t = float(value), 0.0, 0.0, 0

    def step(self, g, lr, wd, beta1=0.9, beta2=0.999, eps=1e-7):
        self.t += 1
        self.m = beta1 * self.m + (1 - beta1) * g
        self.v = beta2 * self.v + (1 - beta2) * g * g
        lr_t = lr * math.sqrt(1 - beta2 ** self.t) / (1 - beta1 ** self.t)
        self.value = self.value - wd * self.value - lr_t * self.m / (math.sqrt(self.v) + eps)
        return self.value


def _fine_tuner_guards(trainer, full_model):
    if getattr(full_model, "pop_prior", None) is not None:
        # train.py:318-320 always adds smoothness_loss, which splits the 8- (or 4 + 4 M-) channel 'predictions' of the population
        # prior into 4 groups and fails on the range division (model.py:729-739): the reference cannot fine-tune
        # with it; the evaluation side (kl_loss, FineTuner.elbo) is built
        raise NotImplementedError("fine-tuning with use_population_prior: the reference's own smoothness_loss "
                                  "raises a shape error on its 8-channel predictions (model.py:729-739)")
    homo = getattr(full_model, "log_sigma", None) is not None
    # heteroscedastic_noise=False: the encoder's sigma head is outside the trained graph (gradient None: no update,
    # no decay); one scalar log sigma is optimised instead
    ranges = None
    if homo:
        w = full_model.encoder_model.weights
        ranges = w.param_ranges([n for n in w.NAMES if n not in ("Ws", "bs")])
    return (ScalarAdamW(full_model.log_sigma) if homo else None), ranges


def prepare_synthetic_dataset(x, y):
    """train.prepare_synthetic_dataset (train.py:82-104): last 10 % of the (already shuffled)
    voxels for validation; batches of 512 'images' of 10x10x5 voxels = 256,000 voxels."""
    vox_per_example = 10 * 10 * 5
    n_examples = x.shape[0] // vox_per_example
    n_valid = n_examples // 10
    n_train_vox = (n_examples - n_valid) * vox_per_example
    train = (x[:n_train_vox], y[:n_train_vox])
    valid = (x[n_train_vox:n_examples * vox_per_example], y[n_train_vox:n_examples * vox_per_example])
    return train, valid, 512 * vox_per_example


def create_and_train_on_synthetic_data(config_dict, params, log=None, sample_size=None, device=None,
                                       max_steps=None):
    model, inner_model, trainer = create_encoder_model(config_dict, params, device=device)
    rank, world, _ = qd.init_from_env()
    log = log or MetricsLog(rank=rank)
    x, y = create_synthetic_dataset(params, _get(config_dict, "full_model"), _get(config_dict, "use_blood"),
                                    _get(config_dict, "misalign_prob"),
                                    uniform_prop=_get(config_dict, "uniform_prop"),
                                    sample_size=sample_size, device=device)
    (tx, ty), (vx, vy), batch = prepare_synthetic_dataset(x, y)
    state = TrainState(trainer.context, model.weights)
    # only the tensors stream 1 runs through receive a gradient here; the stream-2-only ones (residual and
    # gating convolutions, sigma head) are neither updated nor decayed (their Keras gradient is None)
    pt_ranges = model.weights.param_ranges(model.weights.STREAM1_TENSORS)
    lr = float(_get(config_dict, "pt_lr"))
    # AdamW(weight_decay=pt_adamw_decay) wrapped in SWA whose averages are never swapped in
    # (train.py:382-385); plain Adam without decay when use_swa is off.
    wd = float(_get(config_dict, "pt_adamw_decay")) if _get(config_dict, "use_swa") else 0.0
    g = torch.Generator(device=tx.device)
    g.manual_seed(1)
    n = tx.shape[0]
    steps = 0
    # infer_inv_gamma (diagonal family): the four hyper-parameters are Keras variables of the pre-training model, so
    # the same AdamW updates (and decays) them; four scalars, optimised on the host in float64
    hyper = HyperPriorState(model) if trainer._infer_inv_gamma else None
    ig_a = float(_get(config_dict, "inv_gamma_alpha", 0.0) or 0.0)   # train.py:131-135
    ig_b = float(_get(config_dict, "inv_gamma_beta", 0.0) or 0.0)
    use_r2p = bool(_get(config_dict, "use_r2p_loss", False))       # train.py:125, 388
    for epoch in range(int(_get(config_dict, "no_pt_epochs"))):
        perm = torch.randperm(n, generator=g, device=tx.device)
        losses = []
        for b0 in range(0, n, batch):
            idx = perm[b0:b0 + batch]
            a, b = qd.shard_range(idx.numel(), rank, world)
            xb, yb = tx[idx[a:b]], ty[idx[a:b]]
            q1, _ = state.forward(xb, 1)
            lv, gq = state.synth_loss_bwd(yb, q1, 0.0 if hyper else ig_a, 0.0 if hyper else ig_b)
            if hyper:   # model.py:493-507: the learned prior replaces the fixed one
                stats = trainer.context.hyper_prior_bwd(q1, model.hyper_params(), scale=1.0 / q1.shape[0], g_q=gq, loss_v=lv)
                qd.allreduce_sums(stats)
            if use_r2p:   # model.py:475-490; fresh draws every step, keyed by the step and the global voxel
                trainer.context.r2p_loss_bwd(yb, q1, R2P_LOSS_SAMPLES, seed=steps + 1, voxel0=int(b0 + a),
                                             scale=1.0 / q1.shape[0], g_q=gq, loss_v=lv)
            if not trainer._use_mvg:   # logit_gaussian_log_prob (model.py:406-421): no log 2 pi, no Cholesky term
                gq[:, 4] = 0.0
                lv = lv - 1.8378770664093453
            state.backward(1, gq)
            qd.allreduce_mean_(state.grad)
            state.adamw(lr, wd, 0.9, 0.999, 1e-7, ranges=pt_ranges)
            if hyper:
                hyper.step(stats.cpu().numpy(), idx.numel(), lr, wd)
            losses.append(lv.mean())
            steps += 1
            if max_steps and steps >= max_steps:
                break
        loss = float(torch.stack(losses).mean())
        _check_finite(loss, "pre-training loss")
        out1 = model.predict(vx, want=("out1",))[0] if vx.shape[0] else None
        metrics = {"epoch": epoch, "loss": loss}
        if out1 is not None:
            metrics["val_loss"] = float(trainer.synthetic_data_loss(vy, out1, use_r2p, ig_a, ig_b))
            metrics["val_oef_metric"] = float(trainer.oef_metric(vy, out1))
            metrics["val_dbv_metric"] = float(trainer.dbv_metric(vy, out1))
            metrics["val_r2p_metric"] = float(trainer.r2p_metric(vy, out1))
        log.log(metrics)
        if max_steps and steps >= max_steps:
            break
    return model, trainer, inner_model


# ----------------------------------------------------------------------------------------------
# phase 2: ELBO fine-tuning (train.py:285-376)
# ----------------------------------------------------------------------------------------------
def _pad5(t):
    from .model import _pad5 as pad
    return pad(t).contiguous()


def _elbo_bwd(trainer, x, mask, q, prior5, ls, S, K, seed, voxel0):
    """Head gradients of m nll + [m > 0] kl.  Diagonal family (use_mvg=False): the sampled KL is
    replaced by the closed form (model.py:686-716), whose gradient reaches every q parameter, and the
    unused Cholesky column carries no gradient."""
    K = trainer.kl_draws(K, S)   # K per copy of the reference's S-fold tiled batch (model.py:245-246, 656)
    ctx = trainer.context
    if trainer._use_mvg:
        sums, gq, gls, _ = ctx.elbo_bwd(x, mask, q, prior5, ls, S, K, seed=seed, voxel0=voxel0)
        return sums, gq, gls
    sums, gq, gls, _ = ctx.elbo_bwd(x, mask, q, prior5, ls, S, 0, seed=seed, voxel0=voxel0)
    gq[:, 4] = 0.0
    ksums, _ = ctx.kl_diag(q, prior5, mask, g_q=gq, per_voxel=False)
    sums[1] = ksums[1]
    return sums, gq, gls


def prepare_voxel_dataset(data, mask, model):
    """Voxel-batch counterpart of train.prepare_dataset (train.py:17-72): data are masked, the
    stream-1 output of the (pre-trained) model is the per-voxel prior."""
    x = (data * mask[:, None]).contiguous()
    prior = _pad5(model.predict(x, want=("out1",))[0])
    return x, mask.contiguous(), prior


class CropDataset:
    """train.prepare_dataset (train.py:17-72) for image volumes [subj, X, Y, Z, T+1] (mask last):
    optional blank crop [17:-17, 10:-10], data masked, prior = stream-1 prediction per voxel, random
    crop_size x crop_size crops over (X, Y) keeping every slice and channel."""

    def __init__(self, real_data, model, crop_size=20, training=True, blank_crop=True):
        # train.py:18-20 crops [17:-17, 10:-10] unconditionally.  Deviation, documented in DESIGN.md: a volume of at
        # most 34 x 20 in-plane voxels would be left EMPTY by that slice (the reference then fails inside Keras); such
        # volumes -- the small stand-ins of the tests -- are kept whole.  Every volume the reference can process is
        # cropped exactly as it crops it.
        if blank_crop and real_data.shape[1] > 34 and real_data.shape[2] > 20:
            real_data = real_data[:, 17:-17, 10:-10]
        self.mask = real_data[..., -1].contiguous()
        self.data = (real_data[..., :-1] * real_data[..., -1:]).contiguous()
        self.crop = [min(crop_size, self.data.shape[1]), min(crop_size, self.data.shape[2])]
        self.prior = _pad5(model.predict(self.data, want=("out1",))[0])
        self.batch = 38 if training else 3  # train.py:66-70
        self.training = training
        self._cursor = 0

    def next_batch(self, generator):
        n_subj, X, Y = self.data.shape[:3]
        cx, cy = self.crop
        dev = self.data.device
        if self.training:  # shuffled, repeated
            subj = torch.randint(0, n_subj, (self.batch,), generator=generator, device=dev)
        else:
            subj = (torch.arange(self.batch, device=dev) + self._cursor) % n_subj
            self._cursor += self.batch
        x0 = torch.randint(0, X - cx + 1, (self.batch,), generator=generator, device=dev)
        y0 = torch.randint(0, Y - cy + 1, (self.batch,), generator=generator, device=dev)
        ix = (x0[:, None] + torch.arange(cx, device=dev)[None])[:, :, None]
        iy = (y0[:, None] + torch.arange(cy, device=dev)[None])[:, None, :]
        sb = subj[:, None, None]
        return (self.data[sb, ix, iy].contiguous(), self.mask[sb, ix, iy].contiguous(),
                self.prior[sb, ix, iy].contiguous())


def _train_full_model_crops(config_dict, trainer, full_model, study_dataset, train_dataset, log,
                            steps_per_epoch, kl_samples, max_steps):
    rank, world, _ = qd.init_from_env()
    ctx = trainer.context
    model = full_model.encoder_model
    state = TrainState(ctx, model.weights)
    S = trainer._no_samples
    sw = float(_get(config_dict, "smoothness_weight", 0.0) or 0.0)
    decay0, lr0 = float(_get(config_dict, "adamw_decay")), float(_get(config_dict, "ft_lr"))
    beta2 = 0.9 if decay0 > 0.0 else 0.999
    dev = train_dataset.data.device
    g = torch.Generator(device=dev)
    g.manual_seed(2 + rank)   # every rank draws its own crops (data-parallel over crops)
    gv = torch.Generator(device=dev)
    gv.manual_seed(3)
    step = 0
    sigma_opt, ranges = _fine_tuner_guards(trainer, full_model)
    nxt = train_dataset.next_batch(g)
    for epoch in range(int(_get(config_dict, "no_ft_epochs"))):
        tot = torch.zeros(4, dtype=torch.float64, device=dev)
        for _ in range(steps_per_epoch):
            x5, m5, p5 = nxt
            n = m5.numel()
            q, ls = state.forward_spatial(x5)
            if sigma_opt:
                ls = torch.full_like(ls, sigma_opt.value)
            sums, gq, gls = _elbo_bwd(trainer, x5.reshape(n, -1), m5.reshape(n), q, p5.reshape(n, 5), ls, S,
                                      kl_samples, 1000 + step, rank * n)
            tv = ctx.smoothness(q.reshape(m5.shape + (5,)), m5, weight=sw, g_q=gq)
            red = torch.cat([sums, tv] + ([gls.sum(dtype=torch.float64).reshape(1)] if sigma_opt else []))
            qd.allreduce_(red, "allreduce_sums")
            g_sigma = float(red[4] / red[2]) if sigma_opt else None
            red = red[:4]
            state.backward_spatial(gq, None if sigma_opt else gls, red[:3].contiguous())
            # the gradient all-reduce runs on RCCL's stream while this rank draws and gathers its next crops
            work = qd.allreduce_grad_(state.grad, async_op=True)
            nxt = train_dataset.next_batch(g)
            if work is not None:
                work.wait()
            wd = lr_schedule(decay0, step, steps_per_epoch) if decay0 > 0.0 else 0.0
            state.adamw(lr_schedule(lr0, step, steps_per_epoch), wd, 0.9, beta2, 1e-7, ranges=ranges)
            if sigma_opt:
                full_model.log_sigma = sigma_opt.step(g_sigma, lr_schedule(lr0, step, steps_per_epoch), wd, 0.9, beta2)
            tot += red
            step += 1
            if max_steps and step >= max_steps:
                break
        nll, kl, smooth = float(tot[0] / tot[2]), float(tot[1] / tot[2]), float(tot[3] / tot[2])
        _check_finite(nll + kl + smooth, "fine-tuning loss")
        metrics = {"epoch": epoch, "loss": nll + kl + sw * smooth, "predicted_images_loss": nll,
                   "predictions_loss": kl + sw * smooth, "predictions_smoothness_metric": smooth,
                   "predictions_kl_metric": kl}
        # ELBOCallback (train.py:329-357): 4 validation batches, NLL averaged over 10 stochastic passes
        vn = vk = vs = 0.0
        for b in range(4):
            x5, m5, p5 = study_dataset.next_batch(gv)
            nll_b = 0.0
            for i in range(10):
                out = full_model.elbo(x5, m5, p5, kl_samples=kl_samples, seed=7000 + 100 * epoch + 10 * b + i)
                nll_b += float(out["nll"])
            vn += nll_b / 10.0
            vk += float(out["kl"])
            vs += float(ctx.smoothness(_pad5(out["q"]).reshape(m5.shape + (5,)), m5)[0] / m5.sum())
        vn, vk, vs = vn / 4, vk / 4, vs / 4
        metrics.update({"val_nll": vn, "val_elbo": vn + vk, "val_elbo_smooth": vn + vk * 1.0 + vs * sw,
                        "val_smoothness": vs, "val_smoothness_scaled": vs * sw, "val_kl": vk})
        # evaluations whose split-f16 encoder left its operand range and were recomputed on the exact-float32
        # path (ops.Context.vi_fwd, range_check): cumulative count, 0 in a healthy run
        metrics["range_fallbacks"] = int(getattr(trainer.context, "range_fallbacks", 0))
        log.log(metrics)
        if max_steps and step >= max_steps:
            break
    return model


def train_full_model(config_dict, trainer, full_model, study_dataset, train_dataset, log=None,
                     steps_per_epoch=100, batch_voxels=38 * 25 * 25 * 8, kl_samples=70, max_steps=None):
    """Fine-tune stream 2 by minimising nll + 1.0 * kl (+ smoothness_weight * TV, identically 0 on
    voxel batches); kl_weight from the YAML is unused in the reference as well (train.py:313,319)."""
    assert isinstance(trainer, EncoderTrainer)
    rank, world, _ = qd.init_from_env()
    log = log or MetricsLog(rank=rank)
    if isinstance(train_dataset, CropDataset):
        return _train_full_model_crops(config_dict, trainer, full_model, study_dataset, train_dataset, log,
                                       steps_per_epoch, kl_samples, max_steps)
    ctx = trainer.context
    model = full_model.encoder_model
    state = TrainState(ctx, model.weights)
    x, mask, prior = train_dataset
    vx, vmask, vprior = study_dataset
    S = trainer._no_samples
    decay0, lr0 = float(_get(config_dict, "adamw_decay")), float(_get(config_dict, "ft_lr"))
    beta2 = 0.9 if decay0 > 0.0 else 0.999  # AdamW(..., beta_2=0.9) vs plain Adam (train.py:308-312)
    n = x.shape[0]
    g = torch.Generator(device=x.device)
    g.manual_seed(2)
    step = 0

    def draw_batch():
        idx = torch.randint(0, n, (min(batch_voxels, n),), generator=g, device=x.device)
        a, b = qd.shard_range(idx.numel(), rank, world)
        sel = idx[a:b]
        return a, x[sel], mask[sel], prior[sel]

    sigma_opt, ranges = _fine_tuner_guards(trainer, full_model)
    nxt = draw_batch()
    for epoch in range(int(_get(config_dict, "no_ft_epochs"))):
        tot = torch.zeros(3, dtype=torch.float64, device=x.device)
        for _ in range(steps_per_epoch):
            a, xb, mb, pb = nxt
            q2, ls = state.forward(xb, 2)
            if sigma_opt:
                ls = torch.full_like(ls, sigma_opt.value)
            sums, gq, gls = _elbo_bwd(trainer, xb, mb, q2, pb, ls, S, kl_samples, 1000 + step, a)
            if sigma_opt:   # d loss / d log sigma = sum over voxels and taus of the per-tau gradients / sum(mask)
                sums = torch.cat([sums, gls.sum(dtype=torch.float64).reshape(1)])
            qd.allreduce_sums(sums)          # global sum(mask) before the gradient is normalised
            if sigma_opt:
                g_sigma, sums = float(sums[3] / sums[2]), sums[:3].contiguous()
            state.backward(2, gq, None if sigma_opt else gls, sums)
            # shard gradients add up (same 1/sum(m)); the all-reduce runs on RCCL's stream while this rank
            # draws and gathers its next batch (the next forward itself needs the updated weights)
            work = qd.allreduce_grad_(state.grad, async_op=True)
            nxt = draw_batch()
            if work is not None:
                work.wait()
            wd = lr_schedule(decay0, step, steps_per_epoch) if decay0 > 0.0 else 0.0
            state.adamw(lr_schedule(lr0, step, steps_per_epoch), wd, 0.9, beta2, 1e-7, ranges=ranges)
            if sigma_opt:
                full_model.log_sigma = sigma_opt.step(g_sigma, lr_schedule(lr0, step, steps_per_epoch), wd, 0.9, beta2)
            tot += sums
            step += 1
            if max_steps and step >= max_steps:
                break
        nll, kl = float(tot[0] / tot[2]), float(tot[1] / tot[2])
        _check_finite(nll + kl, "fine-tuning loss")
        metrics = {"epoch": epoch, "loss": nll + kl, "predicted_images_loss": nll, "predictions_loss": kl}
        metrics.update(validation_elbo(config_dict, trainer, full_model, (vx, vmask, vprior), kl_samples,
                                       seed=epoch))
        metrics["range_fallbacks"] = int(getattr(trainer.context, "range_fallbacks", 0))   # see _train_full_model_crops
        log.log(metrics)
        if max_steps and step >= max_steps:
            break
    return model


def validation_elbo(config_dict, trainer, full_model, study_dataset, kl_samples=70, seed=0):
    """ELBOCallback.on_epoch_end (train.py:329-357): val_nll is the mean of 10 stochastic passes,
    val_kl one 70-sample estimate, val_elbo = nll + kl; the smoothness terms are 0 on voxels."""
    vx, vmask, vprior = study_dataset
    if vx.shape[0] == 0:
        return {}
    a, b = qd.shard_range(vx.shape[0], *qd.init_from_env()[:2])
    nll = torch.zeros((), dtype=torch.float64, device=vx.device)
    for i in range(10):
        out = full_model.elbo(vx[a:b], vmask[a:b], vprior[a:b], kl_samples=kl_samples,
                              seed=7000 + 10 * seed + i, voxel0=a)
        s = qd.allreduce_sums(out["sums"].clone())
        nll = nll + s[0] / s[2]
        if i == 9:
            kl = s[1] / s[2]
    nll = float(nll / 10.0)
    kl = float(kl)
    sw = float(_get(config_dict, "smoothness_weight", 0.0) or 0.0)
    return {"val_nll": nll, "val_elbo": nll + kl, "val_elbo_smooth": nll + kl * 1.0 + 0.0 * sw,
            "val_smoothness": 0.0, "val_smoothness_scaled": 0.0, "val_kl": kl}


# ----------------------------------------------------------------------------------------------
# data
# ----------------------------------------------------------------------------------------------
def synthetic_voxel_dataset(params, config_dict, n, device, seed):
    """N i.i.d. synthetic voxels: the create_synthetic_dataset priors with
    uniform_prop, forward model + reference noise, mask = 1."""
    rng = np.random.default_rng(seed)
    up = float(_get(config_dict, "uniform_prop", 0.0) or 0.0)
    n_u = round(n * up)

    def draw(lo, hi, mean, std, trunc):
        u = rng.uniform(lo, hi, n_u)
        z = rng.standard_normal(n - n_u) * std + mean
        if trunc:
            bad = (z < lo) | (z > hi)
            while bad.any():
                z[bad] = rng.standard_normal(int(bad.sum())) * std + mean
                bad = (z < lo) | (z > hi)
        else:
            z = np.clip(z, lo, hi)
        out = np.concatenate([u, z])
        rng.shuffle(out)
        return out
    oef = draw(float(params['oef_start']), float(params['oef_end']), float(params['oef_mean']),
               float(params['oef_std']), False)
    dbv = draw(float(params['dbv_start']), float(params['dbv_end']), float(params['dbv_mean']),
               float(params['dbv_std']), True)
    layer = SignalGenerationLayer(dict(params, simulate_noise='True'), _get(config_dict, "full_model"),
                                  _get(config_dict, "use_blood"), device=device, seed=seed)
    y = torch.as_tensor(np.stack([oef, dbv], -1), dtype=torch.float32, device=layer.context.device)
    return layer(y), torch.ones(n, dtype=torch.float32, device=y.device), y


def load_real_volumes(directory, names, T, device, use_brain_mask=False):
    """The reference's .npy volumes [subj, X, Y, 8, T+2] = signals + grey-matter mask + brain mask
    (data_preprocessing.py:265-266; train.py:208-221).  Returns [subj, X, Y, Z, T+1] with ONE mask
    as last channel: the grey-matter mask for the loss (train.py:216,219) or the brain mask for
    whole-brain predictions (:215,218).  Files that already carry T+1 channels pass through."""
    vols = []
    for name in names:
        arr = np.load(os.path.join(directory, name)).astype(np.float32)
        if arr.shape[-1] == T + 2:
            keep = -1 if use_brain_mask else -2
            arr = np.concatenate([arr[..., :T], arr[..., keep:][..., :1]], -1)
        elif arr.shape[-1] != T + 1:
            raise ValueError(f"{name}: last dimension {arr.shape[-1]} is neither T+1 nor T+2 (T={T})")
        vols.append(arr)
    return torch.as_tensor(np.concatenate(vols, 0), device=device)


# ----------------------------------------------------------------------------------------------
# entry points
# ----------------------------------------------------------------------------------------------
def train_model(config_dict, device=None, log=None, pt_sample_size=None, max_pt_steps=None,
                max_ft_steps=None):
    """train.train_model (train.py:188-282) for voxel batches.  Returns (model, trainer, history)."""
    params = get_params('config')
    rank, world, local_rank = qd.init_from_env()
    if device is None and torch.cuda.is_available():
        device = f"cuda:{local_rank % torch.cuda.device_count()}"   # one rank per GPU on a real node
    save_dir = _get(config_dict, "save_directory") or "."
    log = log or MetricsLog(os.path.join(save_dir, "metrics.jsonl") if rank == 0 else None, rank=rank)
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    pt_path = _existing_weights(save_dir, "pt_model")
    final_path = _existing_weights(save_dir, "final_model")
    if os.path.isfile(pt_path):  # train.py:197-199
        model, inner_model, trainer = create_encoder_model(config_dict, params, device=device)
        model.load_weights(pt_path)
    else:
        model, trainer, inner_model = create_and_train_on_synthetic_data(
            config_dict, params, log=log, sample_size=pt_sample_size, device=device,
            max_steps=max_pt_steps)
        if rank == 0:
            model.save_weights(pt_path)

    n_syn = int(_get(config_dict, "synthetic_voxels", 0) or 0)
    if n_syn > 0:
        x, mask, _ = synthetic_voxel_dataset(params, config_dict, n_syn, trainer.context.device, seed=11)
        vx, vmask, _ = synthetic_voxel_dataset(params, config_dict, max(n_syn // 8, 1024),
                                               trainer.context.device, seed=12)
    else:
        d = _get(config_dict, "d")
        if not d or not os.path.exists(d):
            raise Exception('Real data directory not found')  # train.py:204-205
        T, dev = trainer.context.T, trainer.context.device
        train_vol = load_real_volumes(d, ["ASE_scan.npy", "ASE_INF.npy", "ASE_SUP.npy"], T, dev)
        study_vol = load_real_volumes(d, ["hyperv_ase.npy", "baseline_ase.npy"], T, dev)
        # brain-masked copies for image export and their stream-1 priors (train.py:215-236)
        export = {n: load_real_volumes(d, [f"{n}_ase.npy"], T, dev, use_brain_mask=True)
                  for n in ("baseline", "hyperv")}
        export_priors = {n: model.predict(v[..., :-1] * v[..., -1:], want=("out1",))[0][..., :5]
                         for n, v in export.items()}
        tdirs = {n: os.path.join(d, f"transforms_{n}") for n in export}
        baseline_gm = load_real_volumes(d, ["baseline_ase.npy"], T, dev)
        trainer.estimate_population_param_distribution(model, baseline_gm)   # train.py:242
        if rank == 0:                                                         # train.py:248-251
            for n, v in export.items():
                trainer.save_predictions(model, v, os.path.join(save_dir, f"pt_{n}"),
                                         transform_directory=tdirs[n])
    if n_syn > 0:
        train_dataset = prepare_voxel_dataset(x, mask, model)
        study_dataset = prepare_voxel_dataset(vx, vmask, model)
    else:  # prepare_dataset(train_data, model, crop_size) / (study_data, model, 76, training=False)
        train_dataset = CropDataset(train_vol, model, int(_get(config_dict, "crop_size")), training=True)
        study_dataset = CropDataset(study_vol, model, 76, training=False)
    sig_gen_layer = SignalGenerationLayer(dict(params, simulate_noise='False'),
                                          _get(config_dict, "full_model"), _get(config_dict, "use_blood"),
                                          device=trainer.context.device)
    full_model = trainer.build_fine_tuner(model, sig_gen_layer, None, None)
    if os.path.isfile(final_path):  # train.py:260-261
        model.load_weights(final_path)
    else:
        train_full_model(config_dict, trainer, full_model, study_dataset, train_dataset, log=log,
                         max_steps=max_ft_steps)
        if n_syn == 0:
            trainer.estimate_population_param_distribution(model, baseline_gm)   # train.py:265
        if rank == 0:
            model.save_weights(final_path)
            if n_syn == 0:                                                        # train.py:272-279
                for n, v in export.items():
                    trainer.save_predictions(model, v, os.path.join(save_dir, n), transform_directory=tdirs[n],
                                             use_first_op=False, fine_tuner_model=full_model,
                                             priors=export_priors[n])
    return model, trainer, log.history


class ModelBuilder:
    """qbold_build_model.ModelBuilder (:17-82): builds the encoder and loads whichever weights exist."""

    def __init__(self, config_dict, system_params=None, device=None):
        self.config_dict = config_dict
        self.system_params = system_params if system_params else ModelBuilder.get_params()
        model, inner_model, trainer = create_encoder_model(config_dict, self.system_params, device=device)
        self.model, self.inner_model, self.trainer = model, inner_model, trainer
        self.save_dir = os.path.join(os.getcwd(), self.config_dict['save_directory'])
        self.final_model_weights = _existing_weights(self.save_dir, 'final_model')
        self.pt_model_weights = _existing_weights(self.save_dir, 'pt_model')
        self.weight_status = self.load_model_weights()

    @staticmethod
    def get_params():
        return get_params('config')

    def load_model_weights(self):
        if os.path.isfile(self.final_model_weights):
            self.model.load_weights(self.final_model_weights)
            return WeightStatus.FULL_TRAINED
        if os.path.isfile(self.pt_model_weights):
            self.model.load_weights(self.pt_model_weights)
            return WeightStatus.PRE_TRAINED
        print('Model weights do not exit')
        return WeightStatus.NOT_TRAINED


class ModelTrainer(ModelBuilder):
    """qbold_train_model.ModelTrainer (:16-323)."""

    def __init__(self, config_dict, system_params=None, device=None):
        super().__init__(config_dict, system_params, device=device)
        self.history = []

    def train_model(self, **kw):
        t0 = time.time()
        _, _, self.history = train_model(self.config_dict, **kw)
        self.weight_status = WeightStatus.FULL_TRAINED
        return time.time() - t0

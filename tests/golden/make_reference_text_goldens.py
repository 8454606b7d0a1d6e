#!/usr/bin/env python3
"""Runs the REFERENCE'S OWN SOURCE TEXT once and stores inputs + outputs as fixtures.

    python tests/golden/make_reference_text_goldens.py      # build container only: needs /root/reference

signals.py, model.py and logit_mvn.py are imported UNMODIFIED from /root/reference; the TensorFlow / TFP / TFA
modules they import are the NumPy stand-in of tests/golden/tf_standin/, which this script -- and only this
script -- puts on sys.path.  Output: tests/golden/reference_text_goldens.npz (data only: arrays in, arrays out,
plus every random draw the reference code consumed, in call order).

What these fixtures are: a check that the restatement's WIRING is the reference's -- tensor layouts, channel
orders, signs, which mask divides which sum, how the S-fold tiled batch is laid out, which layer feeds which.
What they are NOT: a pin of TensorFlow's arithmetic.  The stand-in computes in NumPy float32 (scipy's J0, NumPy's
reductions), so parity stays "unpinned" in the sense of DESIGN section 2; a misread index or sign, however, is an
O(1) difference and these fixtures catch it.

Nothing under /root/reference travels: only the .npz is committed and shipped to the GPU box.
"""
import configparser
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
if not os.path.isdir(REF):
    raise SystemExit("make_reference_text_goldens.py runs in the build container only (/root/reference is absent)")
sys.path.insert(0, os.path.join(HERE, "tf_standin"))   # for THIS script only
sys.path.insert(0, REF)

import tensorflow as tf   # noqa: E402  (the stand-in)
import logit_mvn as ref_logit_mvn   # noqa: E402  (reference text)
import model as ref_model           # noqa: E402  (reference text)
import signals as ref_signals       # noqa: E402  (reference text)

assert tf.__file__.startswith(HERE), "the stand-in must be the tensorflow that was imported"
assert ref_model.__file__.startswith(REF) and ref_signals.__file__.startswith(REF)

G = {}


def put(case, **arrays):
    for k, v in arrays.items():
        a = v.a if isinstance(v, tf.Tensor) else np.asarray(v)
        assert a.dtype != object, (case, k)
        G[f"{case}/{k}"] = a


def T_(a):
    return tf.Tensor(np.asarray(a, np.float32))


def draws(kind=None):
    out = [a for k, a in tf.random.log if kind is None or k == kind]
    return out


def params(**over):
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "config"))
    p = dict(cfg["DEFAULT"])
    p["simulate_noise"] = "False"
    p.update(over)
    return p


rng = np.random.default_rng(20260401)


def oef_dbv(n):
    y = np.stack([rng.uniform(0.04, 0.84, n), rng.uniform(0.001, 0.201, n)], -1).astype(np.float32)
    y[:4] = [[0.4, 0.12], [0.4, 0.025], [0.04, 0.001], [0.84, 0.201]]
    return y


def q_params(n, scale=1.0):
    """Raw head outputs [mu_oef, raw_s_oef, mu_dbv, raw_s_dbv, raw_c] around the values the network produces."""
    q = np.stack([rng.normal(-0.3, 0.8, n), rng.normal(-0.4, 0.5, n), rng.normal(-1.6, 0.8, n),
                  rng.normal(-0.3, 0.5, n), rng.normal(0.0, 0.8, n)], -1) * scale
    return q.astype(np.float32)


# ----------------------------------------------------------------------------------------------------
# signals.py: SignalGenerationLayer.call / calc_tissue / calc_blood / create_synthetic_dataset
# ----------------------------------------------------------------------------------------------------
y = oef_dbv(256)
for name, full, blood in (("full_blood", True, True), ("full_noblood", True, False),
                          ("loglinear_blood", False, True), ("loglinear_noblood", False, False)):
    lay = ref_signals.SignalGenerationLayer(params(), full, blood)
    put(f"signal/{name}", oef_dbv=y, signal=lay(T_(y)), taus=lay._taus)
lay = ref_signals.SignalGenerationLayer(params(), True, True)
y5 = oef_dbv(2 * 3 * 2 * 2).reshape(2, 3, 2, 2, 2)
put("signal/shape5d", oef_dbv=y5, signal=lay(T_(y5)))
put("signal/dw_r2p", oef_dbv=y, dw=lay.calculate_dw(T_(y[:, 0]), lay.hct), r2p=lay.calculate_r2p(T_(y[:, 0]), T_(y[:, 1]), lay.hct))
lay24 = ref_signals.SignalGenerationLayer(params(tau_start="-0.028", tau_end="0.065", tau_step="0.004"), True, True)
put("signal/protocol24", oef_dbv=y[:128], signal=lay24(T_(y[:128])), taus=lay24._taus)
layh = ref_signals.SignalGenerationLayer(params(), True, True, variable_hct=True)
yh = np.concatenate([y[:128], rng.uniform(0.25, 0.5, (128, 1)).astype(np.float32)], -1)
put("signal/variable_hct", oef_dbv_hct=yh, signal=layh(T_(yh)))
# misalignment augmentation (signals.py:80-96): draws recorded in call order
tf.random.set_seed(11)
laym = ref_signals.SignalGenerationLayer(params(), True, True, misaligned_prob=0.5)
sm = laym(T_(y[:128]))
u_mis, u_idx = draws("uniform")
z_oef, z_dbv = draws("normal")
put("signal/misaligned", oef_dbv=y[:128], signal=sm, u_misaligned=u_mis, from_index=u_idx, z_oef=z_oef, z_dbv=z_dbv,
    prob=np.float32(0.5))
# noise model (signals.py:116-128), T = 11
tf.random.set_seed(12)
layn = ref_signals.SignalGenerationLayer(params(simulate_noise="True"), True, True)
sn = layn(T_(y))
put("signal/noise11", oef_dbv=y, signal=sn, snr_uniform=draws("uniform")[0], z=draws("normal")[0])
# create_synthetic_dataset (signals.py:251-300), noise off, 30 x 30 grid
tf.random.set_seed(13)
xs, ys = ref_signals.create_synthetic_dataset(params(sample_size="30"), True, True, 0.0, uniform_prop=0.1)
u = draws("uniform")
put("synthetic_dataset", x=xs, y=ys, oef_uniform=u[0], oef_normal=draws("normal")[0], dbv_uniform=u[1],
    dbv_truncnorm=draws("truncated_normal")[0], permutation=draws("permutation")[0].astype(np.int64))

# ----------------------------------------------------------------------------------------------------
# model.py / logit_mvn.py: transforms, reparameterisation, densities, losses
# ----------------------------------------------------------------------------------------------------
P = params()
N = 192
et = ref_model.EncoderTrainer(P, no_intermediate_layers=2, no_units=12, activation_type="relu", student_t_df=200,
                              initial_im_sigma=0.05, multi_image_normalisation=False, channelwise_gating=True,
                              use_mvg=True, use_population_prior=False, no_samples=1, heteroscedastic_noise=True,
                              predict_log_data=False)
lm = ref_logit_mvn.LogitMVN()
raw = rng.normal(0, 1.2, (N, 2)).astype(np.float32)
put("transforms", raw=raw, transform_std=et.transform_std(T_(raw)), transform_offdiag=et.transform_offdiag(T_(raw)),
    inv_transform_std=et.inv_transform_std(et.transform_std(T_(raw))), forward_transform=et.forward_transform(T_(raw)),
    oef_dbv=y[:N], backwards_transform=et.backwards_transform(T_(y[:N]), False),
    backwards_transform_logit=et.backwards_transform(T_(y[:N]), True),
    lm_transform_std=lm.transform_std(T_(raw)), lm_transform_offdiag=lm.transform_offdiag(T_(raw)),
    lm_forward_transform=lm.forward_transform(T_(raw)), lm_backwards_transform_logit=lm.backwards_transform(T_(y[:N]), True),
    se_idx=np.int32(et._se_idx))

sig = np.asarray(ref_signals.SignalGenerationLayer(params(), True, True)(T_(y[:N])).a)
noisy = (sig * (1.0 + 0.02 * rng.standard_normal(sig.shape))).astype(np.float32)
clipped = noisy.copy()
clipped[0, 3] = 0.0        # below the clip of normalise_data
et_multi = ref_model.EncoderTrainer(P, multi_image_normalisation=True)
put("normalise", x=clipped, single=et.normalise_data(T_(clipped.reshape(N, 1, 1, 1, 11))),
    multi=et_multi.normalise_data(T_(clipped.reshape(N, 1, 1, 1, 11))))

q = q_params(N)
q5 = q.reshape(N, 1, 1, 1, 5)
ones = np.ones((N, 1, 1, 1, 1), np.float32)
tf.random.set_seed(21)
smp = ref_model.ReparamTrickLayer(et)([T_(q5), T_(ones)])
put("reparam", q=q, z=draws("normal")[0].reshape(N, 2), oef_dbv=smp.a.reshape(N, 2))
tf.random.set_seed(22)
means, stds = et.calculate_means(T_(q5), T_(ones), include_r2p=True, return_stds=True, no_samples=20)
put("moments", q=q, z=np.stack([d.reshape(N, 2) for d in draws("normal")], 1), means=means.a.reshape(N, 3),
    variances=stds.a.reshape(N, 3))

obs = oef_dbv(N)
obs[5] = [0.04, 0.001]   # the clip of the scaled observation at 1e-6
put("logprob", obs=obs, q=q,
    mvg=et.logit_gaussian_mvg_log_prob(T_(obs), T_(q5)).a.reshape(N),
    lm_mvg=lm.logit_gaussian_mvg_log_prob(T_(obs), T_(q5)).a.reshape(N),
    diag=et.logit_gaussian_log_prob(T_(obs[6:]), T_(q5[6:, ..., :4])).a.reshape(N - 6),
    swr=ref_model.EncoderTrainer.squared_whitened_residual(T_(raw), T_(q[:, [0, 2]]), et.transform_std(T_(q[:, 1])),
                                                           et.transform_std(T_(q[:, 3])), et.transform_offdiag(T_(q[:, 4]))),
    lm_swr=ref_logit_mvn.LogitMVN.squared_whitened_residual(T_(raw), T_(q[:, [0, 2]]), lm.transform_std(T_(q[:, 1])),
                                                            lm.transform_std(T_(q[:, 3])), lm.transform_offdiag(T_(q[:, 4]))),
    swr_obs=raw, log_chol_det=ref_model.EncoderTrainer.calculate_log_chol_det(et.transform_std(T_(q[:, 1])),
                                                                               et.transform_std(T_(q[:, 3]))))

# synthetic_data_loss (model.py:449-514)
ytrue = np.concatenate([y[:N], (301.74327 * y[:N, :1] * y[:N, 1:2])], -1).astype(np.float32)
ytrue[:, :2] = np.clip(ytrue[:, :2], [0.05, 0.003], [0.8, 0.195])
ytrue[:, 2] = np.asarray(lay.calculate_r2p(T_(ytrue[:, 0]), T_(ytrue[:, 1]), lay.hct).a)
et_diag = ref_model.EncoderTrainer(P, use_mvg=False, use_population_prior=False)
et_igl = ref_model.EncoderTrainer(P, use_mvg=False, use_population_prior=False, infer_inv_gamma=True)
hyper = np.array([20.0, 2.5, 15.0, 3.0], np.float32)
q8 = np.concatenate([q[:, :4], np.broadcast_to(hyper, (N, 4))], -1).reshape(N, 1, 1, 1, 8)
tf.random.set_seed(23)
loss_r2p = et.synthetic_data_loss(T_(ytrue.reshape(N, 1, 1, 1, 3)), T_(q5), True, 0.0, 0.0)
put("synth_loss", y_true=ytrue, q=q, hyper=hyper,
    mvg=et.synthetic_data_loss(T_(ytrue.reshape(N, 1, 1, 1, 3)), T_(q5), False, 0.0, 0.0),
    mvg_ig=et.synthetic_data_loss(T_(ytrue.reshape(N, 1, 1, 1, 3)), T_(q5), False, 2.0, 0.5),
    diag=et_diag.synthetic_data_loss(T_(ytrue.reshape(N, 1, 1, 1, 3)), T_(q5[..., :4]), False, 0.0, 0.0),
    diag_ig=et_diag.synthetic_data_loss(T_(ytrue.reshape(N, 1, 1, 1, 3)), T_(q5[..., :4]), False, 2.0, 0.5),
    diag_learned_ig=et_igl.synthetic_data_loss(T_(ytrue.reshape(N, 1, 1, 1, 3)), T_(q8), False, 0.0, 0.0),
    mvg_r2p=loss_r2p, r2p_z=np.stack([d.reshape(N, 2) for d in draws("normal")], 1))

# fine_tune_loss_fn (model.py:527-568): y_true = [data, mask], y_pred = [signal, sigma]
mask = (rng.uniform(size=N) > 0.25).astype(np.float32)
pred_y = oef_dbv(N)
pred = np.asarray(ref_signals.SignalGenerationLayer(params(), True, True)(T_(pred_y)).a)
sigma = np.exp(rng.normal(np.log(0.05), 0.3, (N, 11))).astype(np.float32)
data = (noisy * mask[:, None]).astype(np.float32)
y_true = np.concatenate([data, mask[:, None]], -1).reshape(N, 1, 1, 1, 12)
y_pred = np.concatenate([pred, sigma], -1).reshape(N, 1, 1, 1, 22)
put("nll/inputs", data=data, mask=mask, pred=pred, sigma=sigma)
for name, kw in (("gaussian", {}), ("student_t5", dict(student_t_df=5)), ("log_data", dict(predict_log_data=True)),
                 ("multi_image", dict(multi_image_normalisation=True)),
                 ("student_t5_log_multi", dict(student_t_df=5, predict_log_data=True, multi_image_normalisation=True))):
    cfg_kw = dict(student_t_df=200, multi_image_normalisation=False, predict_log_data=False, use_population_prior=False)
    cfg_kw.update(kw)
    e = ref_model.EncoderTrainer(P, **cfg_kw)
    put(f"nll/{name}", mean=e.fine_tune_loss_fn(T_(y_true), T_(y_pred)),
        per_voxel=e.fine_tune_loss_fn(T_(y_true), T_(y_pred), return_mean=False).a.reshape(N))
# S = 3 copies concatenated along the batch (model.py:245-246, 529): three different predictions per voxel
S = 3
preds = [np.asarray(ref_signals.SignalGenerationLayer(params(), True, True)(T_(oef_dbv(N))).a) for _ in range(S)]
y_pred_s = np.concatenate([np.concatenate([p, sigma], -1) for p in preds], 0).reshape(S * N, 1, 1, 1, 22)
e3 = ref_model.EncoderTrainer(P, student_t_df=200, multi_image_normalisation=False, predict_log_data=False, no_samples=S)
put("nll/three_samples", preds=np.stack(preds, 1), mean=e3.fine_tune_loss_fn(T_(y_true), T_(y_pred_s)),
    per_row=e3.fine_tune_loss_fn(T_(y_true), T_(y_pred_s), return_mean=False).a.reshape(S, N))
# homoscedastic sigma (model.py:277-281, 535-537): one scalar in the last channel
eh = ref_model.EncoderTrainer(P, student_t_df=200, multi_image_normalisation=False, predict_log_data=False,
                              heteroscedastic_noise=False)
y_pred_h = np.concatenate([pred, np.full((N, 1), 0.07, np.float32)], -1).reshape(N, 1, 1, 1, 12)
put("nll/homoscedastic", sigma=np.float32(0.07), mean=eh.fine_tune_loss_fn(T_(y_true), T_(y_pred_h)),
    per_voxel=eh.fine_tune_loss_fn(T_(y_true), T_(y_pred_h), return_mean=False).a.reshape(N))

# KL (model.py:592-665): sampled, closed form, diagonal family; true = [prior5, mask]
prior = q_params(N)
true6 = np.concatenate([prior, mask[:, None]], -1).reshape(N, 1, 1, 1, 6)
tf.random.set_seed(31)
kl_mean = et.kl_loss(T_(true6), T_(q5), no_samples=70)
zk = np.stack([d.reshape(N, 2) for d in draws("normal")], 1)
tf.random.set_seed(31)
kl_map = et.kl_loss(T_(true6), T_(q5), return_mean=False, no_samples=70)
put("kl/sampled", q=q, prior=prior, mask=mask, z=zk, mean=kl_mean, per_voxel=kl_map.a.reshape(N),
    closed_form=et.mvg_kl(T_(true6), T_(q5)).a.reshape(N))
true5 = np.concatenate([prior[:, :4], mask[:, None]], -1).reshape(N, 1, 1, 1, 5)
put("kl/diag", mean=et_diag.kl_loss(T_(true5), T_(q5[..., :4])),
    per_voxel=et_diag.kl_loss(T_(true5), T_(q5[..., :4]), return_mean=False).a.reshape(N))
# population prior, diagonal family (model.py:686-716): predictions carry [q4, prior4]; the per-voxel "true" prior is
# ignored except for its mask, and an inverse-gamma(1, 2) cost on the prior's mean log-variances is added
et_pop = ref_model.EncoderTrainer(P, use_mvg=False, use_population_prior=True, mog_components=1)
pop = np.array([-0.97, 0.4, -1.14, 0.6], np.float32)
pred8 = np.concatenate([q[:, :4], np.broadcast_to(pop, (N, 4))], -1).reshape(N, 1, 1, 1, 8)
put("kl/population_diag", pop_prior=pop, mean=et_pop.kl_loss(T_(true5), T_(pred8)),
    per_voxel=et_pop.kl_loss(T_(true5), T_(pred8), return_mean=False).a.reshape(N))

# mixture-of-Gaussians population prior (model.py:666-685): predictions = [q4 | M x prior4]; one reparameterised draw
# per voxel (OEF normal first, then DBV), minus the entropy of q, plus the MEAN of the components' Gaussian NLLs
M = 3
et_mog = ref_model.EncoderTrainer(P, use_mvg=False, use_population_prior=True, mog_components=M)
comps = np.stack([rng.normal(-0.8, 0.6, M), rng.normal(0.2, 0.5, M), rng.normal(-1.2, 0.6, M), rng.normal(0.3, 0.5, M)],
                 -1).astype(np.float32)                                          # [M][4]
pred_mog = np.concatenate([q[:, :4], np.broadcast_to(comps.reshape(-1), (N, 4 * M))], -1).reshape(N, 1, 1, 1, 4 + 4 * M)
tf.random.set_seed(33)
mog_mean = et_mog.kl_loss(T_(true5), T_(pred_mog))
z_mog = np.stack([d.reshape(N) for d in draws("normal")], -1)                  # [N][2]: (z_oef, z_dbv)
tf.random.set_seed(33)
mog_map = et_mog.kl_loss(T_(true5), T_(pred_mog), return_mean=False)
put("kl/mog", components=comps, z=z_mog, mean=mog_mean, per_voxel=mog_map.a.reshape(N))

# smoothness_loss (model.py:726-754) on crops
B, X, Y, Z = 2, 6, 5, 3
qc = q_params(B * X * Y * Z).reshape(B, X, Y, Z, 5)
mc = (rng.uniform(size=(B, X, Y, Z, 1)) > 0.3).astype(np.float32)
truec = np.concatenate([q_params(B * X * Y * Z).reshape(B, X, Y, Z, 5), mc], -1)
put("smoothness", q=qc, mask=mc[..., 0], loss=et.smoothness_loss(T_(truec), T_(qc)),
    diag_loss=et_diag.smoothness_loss(T_(truec[..., [0, 1, 2, 3, 5]]), T_(qc[..., :4])))

# ----------------------------------------------------------------------------------------------------
# create_encoder / build_fine_tuner (model.py:122-286): the two-stream encoder and the ELBO wiring
# ----------------------------------------------------------------------------------------------------
def export_weights(convs, L):
    """Conv3D layers in CREATION order (model.py:181, then per block :144, :152, :156, :164, then :196, :211) ->
    the canonical names of oracle/oracle.py (Keras kernels [kx, ky, kz, in, out])."""
    first, blocks, final, sig_l = convs[0], convs[1:1 + 4 * L], convs[1 + 4 * L], convs[2 + 4 * L]
    w = {"W0": first.kernel[0, 0, 0], "b0": first.bias, "Wf": final.kernel[0, 0, 0], "bf": final.bias,
         "Ws": sig_l.kernel[0, 0, 0], "bs": sig_l.bias}
    for name, off in (("c", 0), ("r1", 1), ("r2", 2), ("g", 3)):
        ks = [blocks[4 * i + off] for i in range(L)]
        w["W" + name] = np.stack([k.kernel if name in ("r1", "r2") else k.kernel[0, 0, 0] for k in ks])
        w["b" + name] = np.stack([k.bias for k in ks])
    return w


def build_encoder(case, U, L, act, channelwise, gate_offset):
    tf.Conv3D.CREATED.clear()
    e = ref_model.EncoderTrainer(P, no_intermediate_layers=L, no_units=U, activation_type=act, student_t_df=200,
                                 initial_im_sigma=0.05, multi_image_normalisation=False,
                                 channelwise_gating=channelwise, use_mvg=True, use_population_prior=False, no_samples=1,
                                 heteroscedastic_noise=True, predict_log_data=False)
    outer, inner = e.create_encoder(gate_offset=gate_offset, resid_init_std=0.05, no_ip_images=11)
    convs = list(tf.Conv3D.CREATED)
    assert len(convs) == 3 + 4 * L
    x0 = noisy[:4].reshape(4, 1, 1, 1, 11)
    outer(T_(x0))                                  # builds the kernels
    for c in convs:                                # non-zero biases: every bias path matters
        c.bias = (c.bias + rng.normal(0, 0.1, c.bias.shape)).astype(np.float32)
    w = export_weights(convs, L)
    put(f"{case}/weights", **w, gate_offset=np.float32(gate_offset), U=np.int32(U), L=np.int32(L),
        channelwise_gating=np.int32(channelwise))
    return e, outer, inner


for case, U, L, act, cw, go in (("encoder_relu", 12, 2, "relu", True, -3.0), ("encoder_gelu", 8, 1, "gelu", True, -1.0),
                                ("encoder_shared_gate", 8, 2, "relu", False, -3.0)):
    e, outer, inner = build_encoder(case, U, L, act, cw, go)
    xv = noisy[:96].reshape(96, 1, 1, 1, 11)
    o1, o2, sg = outer(T_(xv))
    put(f"{case}/voxels", x=noisy[:96], out1=o1.a.reshape(96, 5), out2=o2.a.reshape(96, 5), sigma=sg.a.reshape(96, 11))
    xc = noisy[:2 * 6 * 5 * 2].reshape(2, 6, 5, 2, 11)
    o1, o2, sg = outer(T_(xc))
    put(f"{case}/crops", x=xc, out1=o1, out2=o2, sigma=sg)

# the fine-tuner: encoder -> S copies -> reparameterised sample -> forward model -> [signal, sigma]; then the two
# Keras losses on its outputs (train.py:315-320) with the prior = stream-1 output (train.py:26-31)
S = 2
tf.Conv3D.CREATED.clear()
ef = ref_model.EncoderTrainer(P, no_intermediate_layers=2, no_units=12, activation_type="relu", student_t_df=200,
                              initial_im_sigma=0.05, multi_image_normalisation=False, channelwise_gating=True,
                              use_mvg=True, use_population_prior=False, no_samples=S, heteroscedastic_noise=True,
                              predict_log_data=False)
outer, inner = ef.create_encoder(gate_offset=-3.0, resid_init_std=0.05, no_ip_images=11)
convs = list(tf.Conv3D.CREATED)
Bc, Xc, Yc, Zc = 2, 5, 4, 2
nv = Bc * Xc * Yc * Zc
mask_c = (rng.uniform(size=(Bc, Xc, Yc, Zc, 1)) > 0.2).astype(np.float32)
data_c = noisy[:nv].reshape(Bc, Xc, Yc, Zc, 11) * mask_c        # prepare_dataset masks the data (train.py:54)
prior_c = outer(T_(data_c))[0].a                                 # stream-1 prediction = the per-voxel prior
for c in convs:
    c.bias = (c.bias + rng.normal(0, 0.1, c.bias.shape)).astype(np.float32)
prior_c = outer(T_(data_c))[0].a
sig_layer = ref_signals.SignalGenerationLayer(params(), True, True)
tf.random.set_seed(41)
full = ef.build_fine_tuner(outer, sig_layer, T_(data_c), T_(mask_c))
zs = draws("normal")[0]                                          # [S*B, X, Y, Z, 2]: copy s of voxel v at row s*B + b
outs = full.outputs
y_true_c = np.concatenate([data_c, mask_c], -1)
nll = ef.fine_tune_loss_fn(T_(y_true_c), outs["predicted_images"])
nll_rows = ef.fine_tune_loss_fn(T_(y_true_c), outs["predicted_images"], return_mean=False)
tf.random.set_seed(42)
true_c = np.concatenate([prior_c, mask_c], -1)
kl = ef.kl_loss(T_(true_c), outs["predictions"], no_samples=70)
zk = np.stack(draws("normal"), 0)                                # [K, S*B, X, Y, Z, 2]
tf.random.set_seed(42)
kl_rows = ef.kl_loss(T_(true_c), outs["predictions"], return_mean=False, no_samples=70)
smooth = ef.smoothness_loss(T_(true_c), outs["predictions"])
put("fine_tuner/weights", **export_weights(convs, 2), gate_offset=np.float32(-3.0), U=np.int32(12), L=np.int32(2),
    channelwise_gating=np.int32(1))
put("fine_tuner", data=data_c, mask=mask_c[..., 0], prior=prior_c, S=np.int32(S), zs=zs, zk=zk,
    predictions=outs["predictions"], predicted_images=outs["predicted_images"], nll=nll, kl=kl, smoothness=smooth,
    nll_rows=nll_rows.a.reshape(S * Bc, Xc, Yc, Zc), kl_rows=kl_rows.a.reshape(S * Bc, Xc, Yc, Zc),
    neg_elbo=np.float32(float(nll.a) + float(kl.a)))

# homoscedastic fine-tuner (model.py:277-281): one learned scalar sigma, exp-activated, initial value im_sigma
tf.Conv3D.CREATED.clear()
eh = ref_model.EncoderTrainer(P, no_intermediate_layers=1, no_units=8, activation_type="relu", student_t_df=200,
                              initial_im_sigma=0.08, multi_image_normalisation=False, channelwise_gating=True,
                              use_mvg=True, use_population_prior=False, no_samples=1, heteroscedastic_noise=False,
                              predict_log_data=False)
outer_h, _ = eh.create_encoder(gate_offset=-3.0, resid_init_std=0.05, no_ip_images=11)
convs_h = list(tf.Conv3D.CREATED)
xh = noisy[:64].reshape(64, 1, 1, 1, 11)
mh = np.ones((64, 1, 1, 1, 1), np.float32)
outer_h(T_(xh))
tf.random.set_seed(43)
full_h = eh.build_fine_tuner(outer_h, sig_layer, T_(xh), T_(mh))
put("fine_tuner_homoscedastic/weights", **export_weights(convs_h, 1), gate_offset=np.float32(-3.0), U=np.int32(8),
    L=np.int32(1), channelwise_gating=np.int32(1))
put("fine_tuner_homoscedastic", data=noisy[:64], zs=draws("normal")[0].reshape(64, 2),
    predicted_images=full_h.outputs["predicted_images"].a.reshape(64, 12), initial_im_sigma=np.float32(0.08),
    nll=eh.fine_tune_loss_fn(T_(np.concatenate([xh, mh], -1)), full_h.outputs["predicted_images"]))

out = os.path.join(HERE, "reference_text_goldens.npz")
# use_layer_norm + dropout_rate (model.py:131-140): add_normalizer in front of both activations of the residual path --
# keras Dropout (the models are called outside fit: the identity) and tfa GroupNormalization(groups = 1, axis = -1).
# gamma / beta leave their initialisers (ones / zeros) so that both are seen; two GroupNormalization layers per block,
# in creation order.
import tensorflow_addons as tfa   # noqa: E402  (the stand-in)
for case, U, L, act in (("encoder_layer_norm_relu", 12, 2, "relu"), ("encoder_layer_norm_gelu", 8, 1, "gelu")):
    tf.Conv3D.CREATED.clear()
    tfa.layers.GroupNormalization.CREATED.clear()
    e = ref_model.EncoderTrainer(P, no_intermediate_layers=L, no_units=U, use_layer_norm=True, dropout_rate=0.2,
                                 activation_type=act, student_t_df=200, initial_im_sigma=0.05,
                                 multi_image_normalisation=False, channelwise_gating=True, use_mvg=True,
                                 use_population_prior=False, no_samples=1, heteroscedastic_noise=True,
                                 predict_log_data=False)
    outer, inner = e.create_encoder(gate_offset=-2.0, resid_init_std=0.05, no_ip_images=11)
    convs, norms = list(tf.Conv3D.CREATED), list(tfa.layers.GroupNormalization.CREATED)
    assert len(convs) == 3 + 4 * L and len(norms) == 2 * L
    outer(T_(noisy[:4].reshape(4, 1, 1, 1, 11)))     # builds the kernels and the normalisation parameters
    for c in convs:
        c.bias = (c.bias + rng.normal(0, 0.1, c.bias.shape)).astype(np.float32)
    for g in norms:
        g.gamma = (g.gamma + rng.normal(0, 0.3, g.gamma.shape)).astype(np.float32)
        g.beta = (g.beta + rng.normal(0, 0.2, g.beta.shape)).astype(np.float32)
    ln = np.stack([np.stack([norms[2 * l].gamma, norms[2 * l].beta, norms[2 * l + 1].gamma, norms[2 * l + 1].beta])
                   for l in range(L)])
    put(f"{case}/weights", **export_weights(convs, L), ln=ln, gate_offset=np.float32(-2.0), U=np.int32(U), L=np.int32(L),
        channelwise_gating=np.int32(1))
    xv = noisy[:64].reshape(64, 1, 1, 1, 11)
    o1, o2, sg = outer(T_(xv))
    put(f"{case}/voxels", x=noisy[:64], out1=o1.a.reshape(64, 5), out2=o2.a.reshape(64, 5), sigma=sg.a.reshape(64, 11))
    xc = noisy[:2 * 6 * 5 * 2].reshape(2, 6, 5, 2, 11)
    o1, o2, sg = outer(T_(xc))
    put(f"{case}/crops", x=xc, out1=o1, out2=o2, sigma=sg)

np.savez_compressed(out, **G)
print(f"wrote {out}: {len(G)} arrays, {os.path.getsize(out) / 1024:.0f} KiB")

"""Encoder weight initialisers (host logic; runs once, outside the hot path).

Mirrors the Keras initialisers the reference asks for in EncoderTrainer.create_encoder:
HeNormal for the 1x1x1 layers (model.py:119; Keras' he_normal is a normal truncated at two
standard deviations with stddev sqrt(2 / fan_in) / 0.87962566), RandomNormal(resid_init_std) for
the residual convolutions, the gating convolution and the sigma head (model.py:129,152-165,
211-212), zero biases except the sigma head's constant log(im_loss_sigma) (model.py:213).
The random stream is NumPy's, not TensorFlow's: initial values match in distribution only.
"""
import numpy as np


def weight_shapes(T, U, L, channelwise_gating=True, spatial_taps=1):
    G = U if channelwise_gating else 1
    rs = (L, 3, 3, U, U) if spatial_taps == 9 else (L, U, U)  # Keras (3,3,1,in,out) kernels
    return dict(W0=(T, U), b0=(U,), Wc=(L, U, U), bc=(L, U), Wr1=rs, br1=(L, U),
                Wr2=rs, br2=(L, U), Wg=(L, U, G), bg=(L, G), Wf=(U, 5), bf=(5,),
                Ws=(U, T), bs=(T,))


def _he_normal(rng, shape, fan_in):
    std = np.sqrt(2.0 / fan_in) / 0.87962566103423978
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


def init_encoder_weights(T=11, U=60, L=2, channelwise_gating=True, resid_init_std=0.05,
                         im_loss_sigma=0.05, seed=1, spatial_taps=1, layer_norm=False):
    rng = np.random.default_rng(seed)
    sh = weight_shapes(T, U, L, channelwise_gating, spatial_taps)
    w = {"W0": _he_normal(rng, sh["W0"], T), "Wc": _he_normal(rng, sh["Wc"], U),
         "Wf": _he_normal(rng, sh["Wf"], U)}
    for n in ("Wr1", "Wr2", "Wg", "Ws"):
        w[n] = (rng.standard_normal(sh[n]) * resid_init_std).astype(np.float32)
    for n in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[n] = np.zeros(sh[n], np.float32)
    w["bs"] = np.full(sh["bs"], np.log(im_loss_sigma), np.float32)
    if layer_norm:   # tfa GroupNormalization: gamma 'ones', beta 'zeros'; [L][gamma1, beta1, gamma2, beta2][U]
        w["ln"] = np.tile(np.array([1.0, 0.0, 1.0, 0.0], np.float32)[None, :, None], (L, 1, U))
    return w

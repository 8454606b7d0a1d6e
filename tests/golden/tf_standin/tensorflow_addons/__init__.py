"""Stand-in for `import tensorflow_addons as tfa` (model.py:133).

GroupNormalization follows tfa.layers.GroupNormalization's published algorithm (tensorflow_addons 0.13+,
layers/normalizations.py): the input is reshaped so that the channel axis splits into (groups, channels / groups), mean
and biased variance are taken over every axis but the batch and the group axis (tf.nn.moments), and the result is
tf.nn.batch_normalization with per-channel gamma / beta and epsilon 1e-3.  With groups = 1, axis = -1 (the reference's
call) that is ONE mean and variance per batch element over all positions and channels.  NumPy float32 -- like the rest
of the stand-in it pins where the reference applies the layer, not TensorFlow's arithmetic."""
import types

import numpy as _np

import tensorflow as _tf


def _absent(*a, **k):
    raise NotImplementedError("stand-in: tensorflow_addons optimizers are not implemented")


class GroupNormalization(_tf.keras.layers.Layer):
    CREATED = []

    def __init__(self, groups=32, axis=-1, epsilon=1e-3, center=True, scale=True):
        if axis != -1:
            raise NotImplementedError("stand-in: channel-last only")
        self.groups, self.epsilon, self.center, self.scale = groups, epsilon, center, scale
        self.gamma = self.beta = None
        GroupNormalization.CREATED.append(self)

    def call(self, x):
        a = _np.asarray(_tf._arr(x), _np.float32)
        C = a.shape[-1]
        if self.gamma is None:
            self.gamma, self.beta = _np.ones(C, _np.float32), _np.zeros(C, _np.float32)
        if C // self.groups == 1:
            raise NotImplementedError("stand-in: the instance-norm branch is not on the path")
        g = a.reshape(a.shape[:-1] + (self.groups, C // self.groups))       # [B, ..., groups, C / groups]
        axes = tuple(i for i in range(1, g.ndim) if i != g.ndim - 2)         # all but batch and the group axis
        mean = g.mean(axis=axes, keepdims=True, dtype=_np.float32)
        var = _np.mean(_np.square(g - mean), axis=axes, keepdims=True, dtype=_np.float32)
        shp = (1,) * (g.ndim - 2) + (self.groups, C // self.groups)
        inv = (1.0 / _np.sqrt(var + _np.float32(self.epsilon))).astype(_np.float32) * self.gamma.reshape(shp)
        out = g * inv + (self.beta.reshape(shp) - mean * inv)                # tf.nn.batch_normalization's form
        return _tf.Tensor(out.reshape(a.shape).astype(_np.float32))


layers = types.SimpleNamespace(GroupNormalization=GroupNormalization)
optimizers = types.SimpleNamespace(AdamW=_absent, SWA=_absent)

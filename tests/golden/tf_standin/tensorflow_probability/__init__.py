"""NumPy / SciPy stand-in for the tfp calls of model.py, logit_mvn.py and signals.py (see ../tensorflow).
Densities follow TFP's published formulas, evaluated in float64 and rounded to the operands' dtype."""
import types

import numpy as np
from scipy import special as sp
from scipy import stats

import tensorflow as tf

_T = tf.Tensor


def _a(x):
    return np.asarray(x.a if isinstance(x, _T) else x)


def _f32(x, like=np.float32):
    return _T(np.asarray(x).astype(like))


def clip_by_value_preserve_gradient(x, lo, hi):
    return tf.clip_by_value(x, lo, hi)


class StudentT:
    def __init__(self, df, loc, scale):
        self.df, self.loc, self.scale = (np.asarray(_a(v), np.float64) for v in (df, loc, scale))

    def log_prob(self, x):
        x = np.asarray(_a(x), np.float64)
        y = (x - self.loc) / self.scale
        df = self.df
        lp = (-0.5 * (df + 1.0) * np.log1p(y * y / df) - np.log(np.abs(self.scale)) - 0.5 * np.log(df)
              - 0.5 * np.log(np.pi) - sp.gammaln(0.5 * df) + sp.gammaln(0.5 * (df + 1.0)))
        return _f32(lp)


class InverseGamma:
    def __init__(self, concentration, scale):
        self.c, self.s = (np.asarray(_a(v), np.float64) for v in (concentration, scale))

    def log_prob(self, x):
        x = np.asarray(_a(x), np.float64)
        return _f32(self.c * np.log(self.s) - sp.gammaln(self.c) - (self.c + 1.0) * np.log(x) - self.s / x)


class LogitNormal:
    """TransformedDistribution(Normal(loc, scale), Sigmoid): the KL of two of them is the KL of the normals."""

    def __init__(self, loc, scale):
        self.loc, self.scale = (np.asarray(_a(v), np.float64) for v in (loc, scale))

    def kl_divergence(self, other):
        d = np.log(self.scale) - np.log(other.scale)
        return _f32(0.5 * ((self.loc - other.loc) / other.scale) ** 2 + 0.5 * np.expm1(2.0 * d) - d)


class TruncatedNormal:
    def __init__(self, loc, scale, low, high):
        self.loc, self.scale, self.low, self.high = loc, scale, low, high

    def sample(self, shape):
        a, b = (self.low - self.loc) / self.scale, (self.high - self.loc) / self.scale
        z = stats.truncnorm.rvs(a, b, loc=self.loc, scale=self.scale, size=tuple(shape), random_state=tf.random.rng)
        tf.random.log.append(("truncated_normal", z.astype(np.float32)))
        return _T(z.astype(np.float32))


class VariableLayer(tf.Layer):
    """tfp.layers.VariableLayer: ignores its input and returns activation(variable)."""

    def __init__(self, shape, dtype=None, activation=None, initializer=None):
        self.variable = initializer(tuple(shape))
        self.activation = activation

    def call(self, x):
        v = _T(self.variable)
        return self.activation(v) if self.activation else v


math = types.SimpleNamespace(clip_by_value_preserve_gradient=clip_by_value_preserve_gradient)
distributions = types.SimpleNamespace(StudentT=StudentT, InverseGamma=InverseGamma, LogitNormal=LogitNormal,
                                      TruncatedNormal=TruncatedNormal)
layers = types.SimpleNamespace(VariableLayer=VariableLayer)

"""NumPy stand-in for the handful of `tf.*` / `tf.keras.*` calls on the qBOLD-VI hot path.

NOT TensorFlow, and not product code: test infrastructure, used by
tests/golden/make_reference_text_goldens.py only (which puts this directory on sys.path for that script
alone) so that the reference's own source text -- signals.py, model.py, logit_mvn.py, imported UNMODIFIED
from /root/reference in the build container -- can be executed once and its inputs / outputs stored as
fixtures.  What this pins: the WIRING of the reference (which index, which sign, which tensor goes where).
What it does not pin: TensorFlow's arithmetic (Eigen's j0f, reduction orders, tf.range's rounding), which is
replaced here by NumPy / SciPy float32 -- so fixtures made with it do not turn "parity unpinned" green.

Semantics kept from TensorFlow where the reference relies on them:
  * float32 tensors; a Python / NumPy operand of a binary op is converted to the TENSOR's dtype (so
    `tf.tanh(x) * np.exp(-2.0)` stays float32); two tensors of different dtypes raise, as TF does;
  * eager execution; Keras functional models are evaluated lazily through a small node graph;
  * tf.random.* draws come from one seeded NumPy generator and every draw is RECORDED (`random.log`), so
    that the fixtures carry the exact normals the reference code consumed.
"""
import math as _math
import sys as _sys
import types as _types

import numpy as _np
from scipy import special as _sp

float32 = _np.float32
float64 = _np.float64
int32 = _np.int32
bool_ = _np.bool_
dtypes = _types.SimpleNamespace(float32=float32, float64=float64, int32=int32)


class Tensor:
    __array_ufunc__ = None      # NumPy scalars / arrays on the left defer to the reflected operators below
    __array_priority__ = 1000

    def __init__(self, a, dtype=None):
        if isinstance(a, Tensor):
            a = a.a
        a = _np.asarray(a, dtype=dtype)
        if dtype is None and a.dtype == _np.float64 and not getattr(a, "_keep64", False):
            pass
        self.a = a

    # -- structure ---------------------------------------------------------------------------------
    @property
    def shape(self):
        return TensorShape(self.a.shape)

    @property
    def dtype(self):
        return self.a.dtype.type

    def numpy(self):
        return self.a

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __len__(self):
        return self.a.shape[0]

    def __iter__(self):
        return (Tensor(r) for r in self.a)

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            idx = tuple(_np.asarray(i.a) if isinstance(i, Tensor) else i for i in idx)
        elif isinstance(idx, Tensor):
            idx = idx.a
        return Tensor(self.a[idx])

    def __bool__(self):
        return bool(self.a)

    def __float__(self):
        return float(self.a)

    def __int__(self):
        return int(self.a)

    __index__ = __int__

    def __repr__(self):
        return f"standin.Tensor({self.a!r})"

    # -- arithmetic: the non-tensor operand takes the tensor's dtype (TensorFlow's conversion rule) ---
    def _other(self, o):
        if isinstance(o, Tensor):
            if o.a.dtype != self.a.dtype and self.a.dtype.kind == "f" and o.a.dtype.kind == "f":
                raise TypeError(f"dtype mismatch {self.a.dtype} vs {o.a.dtype} (TensorFlow raises here too)")
            return o.a
        o = _np.asarray(o)
        if o.dtype.kind in "fiu" and self.a.dtype.kind in "fiu":
            o = o.astype(self.a.dtype)
        return o

    def _bin(self, fn, o, reflected=False):
        o = self._other(o)
        with _np.errstate(all="ignore"):
            return Tensor(fn(o, self.a) if reflected else fn(self.a, o))

    def __add__(self, o): return self._bin(_np.add, o)
    def __radd__(self, o): return self._bin(_np.add, o, True)
    def __sub__(self, o): return self._bin(_np.subtract, o)
    def __rsub__(self, o): return self._bin(_np.subtract, o, True)
    def __mul__(self, o): return self._bin(_np.multiply, o)
    def __rmul__(self, o): return self._bin(_np.multiply, o, True)
    def __truediv__(self, o): return self._bin(_np.divide, o)
    def __rtruediv__(self, o): return self._bin(_np.divide, o, True)
    def __pow__(self, o): return self._bin(_np.power, o)
    def __rpow__(self, o): return self._bin(_np.power, o, True)
    def __lt__(self, o): return self._bin(_np.less, o)
    def __le__(self, o): return self._bin(_np.less_equal, o)
    def __gt__(self, o): return self._bin(_np.greater, o)
    def __ge__(self, o): return self._bin(_np.greater_equal, o)
    def __eq__(self, o): return self._bin(_np.equal, o)
    def __ne__(self, o): return self._bin(_np.not_equal, o)
    __hash__ = None
    def __neg__(self): return Tensor(-self.a)
    def __abs__(self): return Tensor(_np.abs(self.a))


class TensorShape(tuple):
    def as_list(self):
        return list(self)


def _arr(x, like=None):
    """ndarray of a tensor / array / Python value; Python floats become float32 as in tf.convert_to_tensor."""
    if isinstance(x, Tensor):
        return x.a
    a = _np.asarray(x)
    if a.dtype == _np.float64 and not isinstance(x, _np.ndarray) and not isinstance(x, _np.generic):
        a = a.astype(_np.float32)
    if like is not None and a.dtype.kind in "fiu" and like.dtype.kind == "f":
        a = a.astype(like.dtype)
    return a


def _shape(s):
    if isinstance(s, Tensor):
        s = s.a
    if isinstance(s, (int, _np.integer)):
        return (int(s),)
    return tuple(int(v.a) if isinstance(v, Tensor) else int(v) for v in s)


def _unary(fn):
    def op(x, name=None):
        with _np.errstate(all="ignore"):
            return Tensor(fn(_arr(x)))
    return op


def convert_to_tensor(x, dtype=None):
    return Tensor(_arr(x) if dtype is None else _np.asarray(_arr(x), dtype=dtype))


def constant(x, dtype=None):
    return convert_to_tensor(x, dtype)


def cast(x, dtype):
    return Tensor(_arr(x).astype(dtype))


def shape(x):
    return Tensor(_np.asarray(_arr(x).shape, dtype=_np.int32))


def reshape(x, shp):
    return Tensor(_arr(x).reshape(_shape(shp)))


def split(x, n, axis=0):
    a = _arr(x)
    if isinstance(n, (list, tuple)):
        return [Tensor(p) for p in _np.split(a, _np.cumsum(n)[:-1], axis=axis)]
    return [Tensor(p) for p in _np.split(a, n, axis=axis)]


def concat(xs, axis):
    arrs = [_arr(x) for x in xs]
    return Tensor(_np.concatenate(arrs, axis=axis))


def stack(xs, axis=0):
    return Tensor(_np.stack([_arr(x) for x in xs], axis=axis))


def expand_dims(x, axis):
    return Tensor(_np.expand_dims(_arr(x), axis))


def zeros_like(x):
    return Tensor(_np.zeros_like(_arr(x)))


def ones_like(x):
    return Tensor(_np.ones_like(_arr(x)))


def clip_by_value(x, lo, hi):
    a = _arr(x)
    return Tensor(_np.clip(a, _np.asarray(lo, a.dtype), _np.asarray(hi, a.dtype)))


def where(c, a, b):
    a_, b_ = _arr(a), _arr(b)
    return Tensor(_np.where(_arr(c), a_, b_))


def logical_and(a, b):
    return Tensor(_np.logical_and(_arr(a), _arr(b)))


def stop_gradient(x):
    return x


def range(start, limit=None, delta=1, dtype=None):   # noqa: A001  (tf.range)
    """tf.range: `size = ceil(|limit - start| / delta)`, element i = start + i * delta, all in `dtype`
    (the multiply form of TF >= 2.x's RangeOp; an accumulating kernel differs by <= 1 ulp)."""
    if limit is None:
        start, limit = 0, start
    if dtype is None:
        dtype = _np.float32 if any(isinstance(v, float) for v in (start, limit, delta)) else _np.int32
    dt = _np.dtype(dtype).type
    s, l, d = dt(start), dt(limit), dt(delta)
    if _np.dtype(dtype).kind == "f":
        size = int(_math.ceil(abs((float(l) - float(s)) / float(d))))
    else:
        size = max(0, -(-(int(l) - int(s)) // int(d)))
    i = _np.arange(size).astype(dtype)
    return Tensor((s + i * d).astype(dtype))


def linspace(start, stop, num):
    """tf.linspace in the operands' dtype: start + i * (stop - start) / (num - 1), last element = stop."""
    a, b = _arr(start), _arr(stop)
    dt = a.dtype.type
    step = dt((b - a) / dt(num - 1))
    out = (a + _np.arange(num).astype(a.dtype) * step).astype(a.dtype)
    out[-1] = b
    return Tensor(out)


def meshgrid(a, b, indexing="xy"):
    x, y = _np.meshgrid(_arr(a), _arr(b), indexing=indexing)
    return Tensor(x), Tensor(y)


def vectorized_map(fn, elems):
    """Row by row, as the reference's per-voxel `compose` is written."""
    n = len(elems[0])
    rows = [fn(tuple(e[i] for e in elems)) for i in _builtin_range(n)]
    return stack(rows, 0)


_builtin_range = __builtins__["range"] if isinstance(__builtins__, dict) else __builtins__.range


def reduce_sum(x, axis=None, keepdims=False):
    return Tensor(_np.sum(_arr(x), axis=axis, keepdims=keepdims, dtype=_arr(x).dtype))


def reduce_mean(x, axis=None, keepdims=False):
    a = _arr(x)
    return Tensor(_np.mean(a, axis=axis, keepdims=keepdims, dtype=a.dtype))


exp = _unary(_np.exp)
sqrt = _unary(_np.sqrt)
square = _unary(_np.square)
tanh = _unary(_np.tanh)
abs = _unary(_np.abs)   # noqa: A001


def _sigmoid(a):
    return (1.0 / (1.0 + _np.exp(-a.astype(_np.float64)))).astype(a.dtype)


def _j0(a):
    # scipy's double-precision J0 rounded to the tensor's dtype (TensorFlow: Eigen's Cephes j0f)
    return _sp.j0(a.astype(_np.float64)).astype(a.dtype)


def _gelu(a):
    return (0.5 * a.astype(_np.float64) * (1.0 + _sp.erf(a.astype(_np.float64) / _math.sqrt(2.0)))).astype(a.dtype)


math = _types.SimpleNamespace(
    exp=exp, sqrt=sqrt, square=square, tanh=tanh, abs=abs,
    log=_unary(_np.log), atanh=_unary(_np.arctanh), is_finite=_unary(_np.isfinite),
    logical_and=logical_and,
    reduce_std=lambda x, axis=None, keepdims=False: Tensor(_np.std(_arr(x), axis=axis, keepdims=keepdims)),
    special=_types.SimpleNamespace(bessel_j0=_unary(_j0)))
nn = _types.SimpleNamespace(sigmoid=_unary(_sigmoid), relu=_unary(lambda a: _np.maximum(a, 0)), gelu=_unary(_gelu))
sigmoid = nn.sigmoid


# ----------------------------------------------------------------------------------------------------
# tf.random: one seeded generator, every draw recorded
# ----------------------------------------------------------------------------------------------------
class _Random:
    def __init__(self):
        self.set_seed(1)

    def set_seed(self, seed):
        self.rng = _np.random.default_rng(seed)
        self.log = []          # [(kind, ndarray)] in call order

    def normal(self, shape, mean=0.0, stddev=1.0, dtype=float32):
        z = self.rng.standard_normal(_shape(shape)).astype(dtype)
        self.log.append(("normal", z))
        return Tensor(z) * stddev + mean if (stddev != 1.0 or mean != 0.0) else Tensor(z)

    def uniform(self, shape, minval=0.0, maxval=1.0, dtype=float32):
        if _np.dtype(dtype).kind == "i":
            u = self.rng.integers(minval, maxval, _shape(shape)).astype(dtype)
        else:
            u = (self.rng.random(_shape(shape)) * (maxval - minval) + minval).astype(dtype)
        self.log.append(("uniform", u))
        return Tensor(u)

    def shuffle(self, x):
        a = _arr(x)
        p = self.rng.permutation(a.shape[0])
        self.log.append(("permutation", p))
        return Tensor(a[p])


random = _Random()


# ----------------------------------------------------------------------------------------------------
# tf.keras: eager layers + a lazy node graph for the functional API
# ----------------------------------------------------------------------------------------------------
class _Node:
    def __init__(self, fn, parents):
        self.fn, self.parents = fn, parents

    def __iter__(self):
        raise TypeError("symbolic tensor: index the model's output list instead")


def _is_sym(x):
    if isinstance(x, _Node):
        return True
    if isinstance(x, (list, tuple)):
        return any(_is_sym(v) for v in x)
    return False


def _evaluate(node, memo):
    if not isinstance(node, _Node):
        if isinstance(node, (list, tuple)):
            return [_evaluate(v, memo) for v in node]
        return node
    if id(node) not in memo:
        memo[id(node)] = node.fn(*[_evaluate(p, memo) for p in node.parents])
    return memo[id(node)]


class Layer:
    def __init__(self, *a, **k):
        pass

    def __call__(self, inputs, *a, **k):
        if _is_sym(inputs):
            return _Node(lambda v: self.call(v, *a, **k), [inputs])
        return self.call(inputs, *a, **k)


_ACT = {None: lambda x: x, "linear": lambda x: x, "relu": nn.relu, "gelu": nn.gelu}


def _activation(act):
    return act if callable(act) else _ACT[act]


class Activation(Layer):
    def __init__(self, activation):
        self.fn = _activation(activation)

    def call(self, x):
        return self.fn(x)


class Lambda(Layer):
    def __init__(self, fn):
        self.fn = fn

    def call(self, x):
        return self.fn(x)


class Dropout(Layer):
    """keras.layers.Dropout outside model.fit: the identity (the fixtures call the models in inference mode)."""
    def __init__(self, rate):
        self.rate = rate

    def call(self, x, training=False):
        if training:
            raise NotImplementedError("stand-in: training-mode dropout draws from TensorFlow's random stream")
        return x


class Conv3D(Layer):
    """Channel-last 3-D cross-correlation, stride 1; 'valid' for 1x1x1 kernels, 'same' zero padding otherwise.
    Kernel [kx, ky, kz, in, out] (the Keras layout).  CREATED registers instances in creation order."""
    CREATED = []

    def __init__(self, filters, kernel_size, padding="valid", kernel_initializer=None, bias_initializer=None,
                 activation=None):
        self.filters, self.kernel_size, self.padding = filters, tuple(kernel_size), padding
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
        self.act = _activation(activation)
        self.kernel = self.bias = None
        Conv3D.CREATED.append(self)

    def build(self, cin):
        shp = self.kernel_size + (cin, self.filters)
        init = self.kernel_initializer or initializers.GlorotUniform()
        self.kernel = init(shp)
        self.bias = (self.bias_initializer or initializers.Constant(0.0))((self.filters,))

    def call(self, x):
        a = _arr(x)
        if self.kernel is None:
            self.build(a.shape[-1])
        kx, ky, kz = self.kernel_size
        if (kx, ky, kz) != (1, 1, 1) and self.padding != "same":
            raise NotImplementedError
        px, py, pz = kx // 2, ky // 2, kz // 2
        ap = _np.pad(a, ((0, 0), (px, px), (py, py), (pz, pz), (0, 0)))
        X, Y, Z = a.shape[1:4]
        out = _np.zeros(a.shape[:4] + (self.filters,), _np.float32)
        for i in _builtin_range(kx):
            for j in _builtin_range(ky):
                for k in _builtin_range(kz):
                    out += _np.einsum("bxyzc,co->bxyzo", ap[:, i:i + X, j:j + Y, k:k + Z, :],
                                      self.kernel[i, j, k]).astype(_np.float32)
        return self.act(Tensor(out + self.bias))


def Input(shape=None, ragged=False, **k):
    return _Node(None, [])


class Model:
    def __init__(self, inputs, outputs):
        self.inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self.outputs = outputs

    def _run(self, values):
        values = list(values) if isinstance(values, (list, tuple)) else [values]
        memo = {id(n): v for n, v in zip(self.inputs, values)}
        if isinstance(self.outputs, dict):
            return {k: _evaluate(v, memo) for k, v in self.outputs.items()}
        return _evaluate(self.outputs, memo)

    def __call__(self, x):
        if _is_sym(x):
            whole = _Node(lambda v: self._run(v), [x])
            if isinstance(self.outputs, (list, tuple)):
                return [_Node(lambda vals, i=i: vals[i], [whole]) for i in _builtin_range(len(self.outputs))]
            return whole
        return self._run(x)

    def predict(self, x):
        out = self._run(x)
        return [_arr(o) for o in out] if isinstance(out, list) else out


class _Initializers:
    """Seeded NumPy draws with the Keras initialisers' distributions (values are arbitrary test weights)."""
    rng = _np.random.default_rng(1234)

    class HeNormal:
        def __call__(self, shp):
            fan_in = int(_np.prod(shp[:-1]))
            z = _np.clip(_Initializers.rng.standard_normal(shp), -2.0, 2.0)
            return (z * _math.sqrt(2.0 / fan_in) / 0.87962566103423978).astype(_np.float32)

    class GlorotUniform:
        def __call__(self, shp):
            lim = _math.sqrt(6.0 / (int(_np.prod(shp[:-1])) + shp[-1]))
            return _Initializers.rng.uniform(-lim, lim, shp).astype(_np.float32)

    class RandomNormal:
        def __init__(self, mean=0.0, stddev=0.05):
            self.mean, self.stddev = mean, stddev

        def __call__(self, shp):
            return (_Initializers.rng.standard_normal(shp) * self.stddev + self.mean).astype(_np.float32)

    class Constant:
        def __init__(self, value):
            self.value = value

        def __call__(self, shp):
            return _np.broadcast_to(_np.asarray(self.value, _np.float32), shp).copy()

    constant = Constant
    random_normal = RandomNormal


initializers = _Initializers()
layers = _types.SimpleNamespace(Layer=Layer, Conv3D=Conv3D, Lambda=Lambda, Activation=Activation, Dropout=Dropout,
                                Input=Input)
keras = _types.ModuleType("tensorflow.keras")
keras.layers, keras.Model, keras.initializers = layers, Model, initializers
_sys.modules["tensorflow.keras"] = keras

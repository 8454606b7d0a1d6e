"""Keras HDF5 weight files (`model.save_weights('pt_model.h5')`, train.py:201,247,270) <-> the
canonical tensor names of this package.

h5py is NOT a dependency of this package (it is absent from the build image), so this module
imports it lazily and says so when it is missing; `.npz` with the canonical names stays the native
format.  The mapping follows the layout Keras 2.x writes for the reference's model and goes by ORDER
AND SHAPE, not by layer names (Keras numbers `conv3d_<n>` by creation order in the process, so the
names shift when several models were built):

  root.attrs['layer_names']  -> the outer model's layers in order: input, lambda (normalise_data),
      the first 1x1x1 conv, the inner functional model, the sigma-head conv (model.py:176-222)
  group(layer).attrs['weight_names'] -> that layer's variables in creation order; the inner model
      holds, per create_block call (model.py:142-174): shared 1x1x1 conv, 3x3x1 conv, 3x3x1 conv,
      gating conv -- then the final layer (model.py:196)

so the flattened variable list is  W0 b0 | (Wc bc Wr1 br1 Wr2 br2 Wg bg) x L | Wf bf | Ws bs  with
Conv3D kernels shaped [kx, ky, kz, in, out].

UNVERIFIED AGAINST A REAL FILE: the reference ships no weight file and neither TensorFlow nor h5py
can run here; tests exercise this module against an in-memory stand-in for the h5py objects.
"""
import numpy as np

BLOCK = ("Wc", "bc", "Wr1", "br1", "Wr2", "br2", "Wg", "bg")


def _h5py():
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - depends on the user's environment
        raise ImportError("reading or writing Keras .h5 weight files needs h5py (not installed); "
                          "use the .npz files written by save_weights instead") from e
    return h5py


def _names(attr):
    return [n.decode("utf8") if isinstance(n, (bytes, np.bytes_)) else str(n) for n in attr]


def flatten_variables(f):
    """All variables of a Keras weights file (an open h5py.File or anything with the same mapping /
    .attrs interface) in layer order, as float32 arrays."""
    out = []
    for layer in _names(f.attrs["layer_names"]):
        g = f[layer]
        for w in _names(g.attrs.get("weight_names", [])):
            out.append(np.asarray(g[w], dtype=np.float32))
    return out


def _kernel(a, taps):
    """Conv3D kernel [kx, ky, kz, in, out] -> [in, out] (1x1x1) or [3, 3, in, out] (3x3x1)."""
    if a.ndim != 5 or a.shape[2] != 1:
        raise ValueError(f"expected a Conv3D kernel [kx, ky, 1, in, out], got shape {a.shape}")
    if taps == 1:
        if a.shape[:2] != (1, 1):
            raise ValueError(f"expected a 1x1x1 kernel, got {a.shape}")
        return a[0, 0, 0]
    if a.shape[:2] != (3, 3):
        raise ValueError(f"expected a 3x3x1 kernel, got {a.shape}")
    return a[:, :, 0]


def variables_to_canonical(variables):
    """Flattened Keras variables -> dict of canonical arrays (per-block tensors stacked on axis 0)."""
    v = list(variables)
    if len(v) < 6 or (len(v) - 6) % 8 != 0:
        raise ValueError(f"{len(v)} variables do not fit  W0 b0 | 8 per block | Wf bf | Ws bs")
    L = (len(v) - 6) // 8
    w = {"W0": _kernel(v[0], 1), "b0": v[1]}
    T, U = w["W0"].shape
    blocks = {k: [] for k in BLOCK}
    for l in range(L):
        b = v[2 + 8 * l: 10 + 8 * l]
        blocks["Wc"].append(_kernel(b[0], 1)); blocks["bc"].append(b[1])
        blocks["Wr1"].append(_kernel(b[2], 9)); blocks["br1"].append(b[3])
        blocks["Wr2"].append(_kernel(b[4], 9)); blocks["br2"].append(b[5])
        blocks["Wg"].append(_kernel(b[6], 1)); blocks["bg"].append(b[7])
    for k in BLOCK:
        w[k] = np.stack(blocks[k])
    w["Wf"], w["bf"] = _kernel(v[2 + 8 * L], 1), v[3 + 8 * L]
    w["Ws"], w["bs"] = _kernel(v[4 + 8 * L], 1), v[5 + 8 * L]
    ok = (w["Wc"].shape[1:] == (U, U) and w["Wr1"].shape[1:] == (3, 3, U, U) and w["Wf"].shape[0] == U
          and w["Wf"].shape[1] in (4, 5) and w["Ws"].shape == (U, T) and w["Wg"].shape[1] == U
          and w["Wg"].shape[2] in (1, U))
    if not ok:
        raise ValueError("variable shapes do not match the reference encoder "
                         f"(T={T}, U={U}, L={L}): " + ", ".join(f"{k}{tuple(a.shape)}" for k, a in w.items()))
    return w


def load_keras_h5(path):
    """-> canonical weight dict from a Keras `.h5` weights file of the reference's encoder."""
    with _h5py().File(path, "r") as f:
        return variables_to_canonical(flatten_variables(f))


def canonical_to_variables(w):
    """Inverse of variables_to_canonical: [(name, array)] in Keras order, kernels as Conv3D 5-D."""
    k1 = lambda a: np.asarray(a, np.float32)[None, None, None]
    k9 = lambda a: np.asarray(a, np.float32)[:, :, None]
    L = np.asarray(w["Wc"]).shape[0]
    out = [("conv3d/kernel:0", k1(w["W0"])), ("conv3d/bias:0", np.asarray(w["b0"], np.float32))]
    n = 1
    for l in range(L):
        for name, kern in (("Wc", k1), ("Wr1", k9), ("Wr2", k9), ("Wg", k1)):
            out.append((f"conv3d_{n}/kernel:0", kern(w[name][l])))
            out.append((f"conv3d_{n}/bias:0", np.asarray(w["b" + name[1:].lower()][l], np.float32)))
            n += 1
    out += [(f"conv3d_{n}/kernel:0", k1(w["Wf"])), (f"conv3d_{n}/bias:0", np.asarray(w["bf"], np.float32))]
    n += 1
    out += [(f"conv3d_{n}/kernel:0", k1(w["Ws"])), (f"conv3d_{n}/bias:0", np.asarray(w["bs"], np.float32))]
    return out


def save_keras_h5(path, w):
    """Write canonical weights in the layout load_keras_h5 reads (outer layers: conv3d, model, sigma conv)."""
    h5py = _h5py()
    var = canonical_to_variables(w)
    groups = [("conv3d", var[:2]), ("model", var[2:-2]), (var[-2][0].split("/")[0], var[-2:])]
    with h5py.File(path, "w") as f:
        f.attrs["layer_names"] = [g.encode("utf8") for g, _ in groups]
        f.attrs["backend"] = b"tensorflow"
        for gname, items in groups:
            g = f.create_group(gname)
            g.attrs["weight_names"] = [n.encode("utf8") for n, _ in items]
            for n, a in items:
                g.create_dataset(n, data=a)

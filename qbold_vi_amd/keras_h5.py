"""Keras HDF5 weight files (`model.save_weights('pt_model.h5')`, train.py:201,247,270) <-> the
canonical tensor names of this package.

h5py is NOT a dependency of this package (it is absent from the build image), so this module
imports it lazily and says so when it is missing; `.npz` with the canonical names stays the native
format.  The reference creates its Conv3D layers in one fixed order (model.py:122-223: first 1x1x1 conv; per
create_block call the shared 1x1x1 conv, two 3x3x1 convs, the gating conv; then the final layer and the sigma
head) and Keras numbers them `conv3d`, `conv3d_1`, ... in CREATION order, whereas the order in which a
Functional model's variables are WRITTEN follows graph depth (a block's 3x3x1 kernels come before its 1x1x1
ones).  So variables are first paired (kernel, bias) per layer and sorted by the layer's creation number (the
numbers shift by a constant when several models were built in one process: only their order is used); files
whose names carry no number fall back to roles by shape inside each block (3x3x1 kernels are Wr1 then Wr2, of
the 1x1x1 kernels the first is the shared conv, the other the gate).  What Keras 2.x writes:

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


def _named_variables(f):
    out = []
    for layer in _names(f.attrs["layer_names"]):
        g = f[layer]
        for w in _names(g.attrs.get("weight_names", [])):
            out.append((w, np.asarray(g[w], dtype=np.float32)))
    return out


def _creation_number(weight_name):
    """'model/conv3d_7/kernel:0' -> 7, 'conv3d/bias:0' -> 0; None when the layer name carries no number."""
    import re
    parts = weight_name.split("/")
    layer = parts[-2] if len(parts) >= 2 else ""
    m = re.fullmatch(r"(?:.*?)conv3d(?:_(\d+))?", layer)
    if not m:
        return None
    return int(m.group(1)) if m.group(1) else 0


def _split_hyper_prior(named):
    """A checkpoint saved with infer_inv_gamma=True carries the tfp VariableLayer's single rank-1 variable of four
    logs (model.py:201-205) in a layer group of its own: taken out before the Conv3D layers are paired."""
    groups = {}
    for name, a in named:
        groups.setdefault(name.rsplit("/", 1)[0], []).append(a)
    hyper_keys = [k for k, arrs in groups.items() if len(arrs) == 1 and arrs[0].ndim == 1 and arrs[0].shape == (4,)]
    if len(hyper_keys) > 1:
        raise ValueError("more than one candidate for the inverse-gamma hyper-prior variable")
    if not hyper_keys:
        return named, None
    key = hyper_keys[0]
    return [(n, a) for n, a in named if n.rsplit("/", 1)[0] != key], np.asarray(groups[key][0], np.float32)


def _pair_layers(named):
    """[(weight name, array)] in file order -> [(kernel, bias)] per layer, in file order."""
    layers, index = [], {}
    for name, a in named:
        key = name.rsplit("/", 1)[0]
        if key not in index:
            index[key] = len(layers)
            layers.append([key, None, None])
        slot = 1 if a.ndim == 5 else 2
        if layers[index[key]][slot] is not None:
            raise ValueError(f"layer {key!r} has two variables of the same rank")
        layers[index[key]][slot] = a
    for key, k, b in layers:
        if k is None or b is None:
            raise ValueError(f"layer {key!r} does not hold one Conv3D kernel and one bias")
    return layers


def _by_roles(layers):
    """File order (graph depth) -> creation order, by kernel shape inside each block: the outer model's first
    conv and sigma head are the first and last layers; the inner model holds 4 L block layers + the final one."""
    first, last, inner = layers[0], layers[-1], layers[1:-1]
    final = [l for l in inner if l[1].shape[:2] == (1, 1) and l[1].shape[-1] in (4, 5) and l[1].shape[-1] != l[1].shape[-2]]
    if len(final) != 1:
        raise ValueError("cannot tell the final layer from the block layers by shape")
    body = [l for l in inner if l is not final[0]]
    if len(body) % 4:
        raise ValueError(f"{len(body)} block layers are not four per block")
    L = len(body) // 4
    k3 = [l for l in body if l[1].shape[:2] == (3, 3)]
    k1 = [l for l in body if l[1].shape[:2] == (1, 1)]
    if len(k3) != 2 * L or len(k1) != 2 * L:
        raise ValueError("a block needs two 3x3x1 and two 1x1x1 kernels")
    out = [first]
    for l in range(L):
        shared, gate = k1[2 * l], k1[2 * l + 1]
        if shared[1].shape[-1] != shared[1].shape[-2] and gate[1].shape[-1] == gate[1].shape[-2]:
            shared, gate = gate, shared   # shared gating has one output: it cannot be the U -> U conv
        out += [shared, k3[2 * l], k3[2 * l + 1], gate]
    return out + [final[0], last]


def _split_group_norm(named):
    """use_layer_norm=True (model.py:139): tfa GroupNormalization layers hold two rank-1 variables, gamma and beta,
    created in block order -- two layers per block (model.py:150, 154).  Taken out before the Conv3D layers are paired;
    returned as [L][4][U] (gamma1, beta1, gamma2, beta2), the canonical 'ln' tensor."""
    import re
    gn = [(n, a) for n, a in named if "group_normalization" in n]
    if not gn:
        return named, None
    layers = {}
    for n, a in gn:
        key = n.rsplit("/", 1)[0]
        m = re.search(r"group_normalization(?:_(\d+))?$", key)
        layers.setdefault((int(m.group(1)) if m and m.group(1) else 0, key), {})["gamma" if "gamma" in n else "beta"] = a
    ordered = [layers[k] for k in sorted(layers)]
    if len(ordered) % 2 or any(set(d) != {"gamma", "beta"} for d in ordered):
        raise ValueError("GroupNormalization layers do not come as (gamma, beta), two layers per block")
    ln = np.stack([np.stack([ordered[2 * l]["gamma"], ordered[2 * l]["beta"], ordered[2 * l + 1]["gamma"],
                             ordered[2 * l + 1]["beta"]]) for l in range(len(ordered) // 2)]).astype(np.float32)
    return [(n, a) for n, a in named if "group_normalization" not in n], ln


def flatten_variables(f):
    """All variables of a Keras weights file (an open h5py.File or anything with the same mapping /
    .attrs interface) as float32 arrays in the reference's layer CREATION order, kernel before bias."""
    named, ln = _split_group_norm(_named_variables(f))
    named, hyper = _split_hyper_prior(named)
    if ln is not None:
        flatten_variables.last_ln = ln    # picked up by load_keras_h5 (the flattened list keeps its documented shape)
    layers = _pair_layers(named)
    numbers = [_creation_number(key + "/kernel:0") for key, _, _ in layers]
    if all(n is not None for n in numbers) and len(set(numbers)) == len(numbers):
        layers = [l for _, l in sorted(zip(numbers, layers), key=lambda t: t[0])]
    else:
        layers = _by_roles(layers)
    out = []
    for _, k, b in layers:
        out += [k, b]
    if hyper is not None:
        out.append(hyper)       # trailing rank-1 variable: variables_to_canonical maps it to 'hyper_prior'
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
    hyper = None
    if v and np.asarray(v[-1]).ndim == 1 and np.asarray(v[-1]).shape == (4,) and len(v) % 2 == 1:
        hyper = np.asarray(v.pop(), np.float32)      # the infer_inv_gamma VariableLayer (model.py:201-205)
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
    if hyper is not None:
        w["hyper_prior"] = hyper
    return w


def load_keras_h5(path):
    """-> canonical weight dict from a Keras `.h5` weights file of the reference's encoder."""
    with _h5py().File(path, "r") as f:
        flatten_variables.last_ln = None
        w = variables_to_canonical(flatten_variables(f))
        if flatten_variables.last_ln is not None:
            w["ln"] = flatten_variables.last_ln
        return w


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
    if w.get("ln") is not None:              # tfa GroupNormalization layers, two per block: gamma, beta
        ln = np.asarray(w["ln"], np.float32)
        for l in range(ln.shape[0]):
            for j in range(2):
                name = "group_normalization" + ("" if 2 * l + j == 0 else f"_{2 * l + j}")
                out.append((f"{name}/gamma:0", ln[l, 2 * j]))
                out.append((f"{name}/beta:0", ln[l, 2 * j + 1]))
    if w.get("hyper_prior") is not None:     # tfp.layers.VariableLayer names its variable 'constant:0'
        out.append(("variable_layer/constant:0", np.asarray(w["hyper_prior"], np.float32)))
    return out


def save_keras_h5(path, w):
    """Write canonical weights in the layout load_keras_h5 reads (outer layers: conv3d, model, sigma conv)."""
    h5py = _h5py()
    var = canonical_to_variables(w)
    hyper = [var.pop()] if var[-1][0].startswith("variable_layer/") else []
    norms = [v for v in var if v[0].startswith("group_normalization")]
    var = [v for v in var if not v[0].startswith("group_normalization")]
    # the hyper-prior layer is created inside the inner model, after the final layer (model.py:196-205); the
    # GroupNormalization layers belong to the inner model's blocks
    groups = [("conv3d", var[:2]), ("model", var[2:-2] + norms + hyper), (var[-2][0].split("/")[0], var[-2:])]
    with h5py.File(path, "w") as f:
        f.attrs["layer_names"] = [g.encode("utf8") for g, _ in groups]
        f.attrs["backend"] = b"tensorflow"
        for gname, items in groups:
            g = f.create_group(gname)
            g.attrs["weight_names"] = [n.encode("utf8") for n, _ in items]
            for n, a in items:
                g.create_dataset(n, data=a)

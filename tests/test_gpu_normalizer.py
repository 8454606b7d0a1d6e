"""EncoderTrainer's use_layer_norm and dropout_rate (model.py:131-140): tfa GroupNormalization(groups = 1, axis = -1) and
keras Dropout in front of the residual path's two activations, on the layer-wise kernels -- forward against the oracle's
restatement (voxel batches, crops, relu and gelu), the training-mode dropout stream against the oracle's, the backward
(kernel, bias and GroupNormalization-parameter gradients) against central differences of the float64 oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

WEIGHT_NAMES = ("W0", "b0", "Wc", "bc", "Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Wf", "bf", "Ws", "bs")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


@pytest.fixture(scope="module")
def ctx(params):
    from qbold_vi_amd.ops import Context
    c = Context(params, full_model=True, include_blood=True)
    c.set_grad_node0(False)
    return c


@pytest.fixture(scope="module")
def oracle64(params):
    from oracle.oracle import Oracle
    o = Oracle("f64", params, node0_zero=True)
    yield o
    o.lib.qbo_set_node0_zero(0)


def make(ctx, U, L, activation="relu", layer_norm=True, dropout_rate=0.0, seed=3, gate_offset=-1.0):
    from oracle.oracle import init_weights
    from qbold_vi_amd.ops import EncoderWeights
    w = init_weights(T=11, U=U, L=L, channelwise_gating=True, seed=seed, taps=9, resid_init_std=0.08)
    rng = np.random.default_rng(seed)
    for k in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
    w["gate_offset"] = gate_offset
    ln = None
    if layer_norm:   # gamma about 1, beta about 0: away from the initialisers, so that both carry a gradient of their own
        ln = np.stack([np.stack([1.0 + 0.3 * rng.standard_normal(U), 0.2 * rng.standard_normal(U),
                                 1.0 + 0.3 * rng.standard_normal(U), 0.2 * rng.standard_normal(U)]) for _ in range(L)])
        ln = ln.astype(np.float32)
    ew = EncoderWeights(ctx, 11, U, L, True, gate_offset, spatial_taps=9, activation=activation, layer_norm=layer_norm,
                        dropout_rate=dropout_rate)
    ew.set_from_arrays(dict(w, ln=ln) if layer_norm else w)
    return w, ln, ew


def crop_batch(oracle32, B, X, Y, Z, seed):
    from oracle.oracle import synth_inputs
    x, _ = synth_inputs(B * X * Y * Z, seed=seed, oracle=oracle32)
    return x.reshape(B, X, Y, Z, 11)


def with_activation(oracle, activation):
    oracle.lib.qbo_set_activation_gelu(1 if activation == "gelu" else 0)


@pytest.mark.parametrize("activation", ["relu", "gelu"])
@pytest.mark.parametrize("geometry", [(3, 5, 4, 2), (40, 1, 1, 1)])
def test_layer_norm_forward_matches_oracle(ctx, oracle32, activation, geometry):
    """Crops: statistics over a whole crop and its channels; (N,1,1,1,T) voxel batches: over a voxel's channels -- through
    the crop entry point and (voxel batches) the voxel entry point, which sees the centre tap."""
    from qbold_vi_amd.ops import TrainState
    B, X, Y, Z = geometry
    w, ln, ew = make(ctx, 20, 2, activation)
    x = crop_batch(oracle32, B, X, Y, Z, seed=7)
    with_activation(oracle32, activation)
    try:
        q_want, sg_want = oracle32.encoder_fwd_spatial(w, x, ln=ln)
        q_plain, _ = oracle32.encoder_fwd_spatial(w, x)
    finally:
        with_activation(oracle32, "relu")
    st = TrainState(ctx, ew, optimiser_state=False)
    q, ls = st.forward_spatial(dev(x))
    n = B * X * Y * Z
    assert np.max(np.abs(q.cpu().numpy() - q_want.reshape(n, 5))) < 3e-5
    assert np.max(np.abs(np.exp(ls.cpu().numpy()) / sg_want.reshape(n, 11) - 1)) < 1e-4
    assert np.max(np.abs(q_plain - q_want)) > 1e-2          # the normalisation does something
    if X == Y == Z == 1:
        q2, ls2 = st.forward(dev(x.reshape(n, 11)), 2)
        assert np.max(np.abs(q2.cpu().numpy() - q_want.reshape(n, 5))) < 3e-5
        # the reference-shaped API takes the same route (EncoderModel.predict -> layer-wise kernels)
        o1, o2, sg = ctx.encoder_fwd(ew, dev(x.reshape(n, 11)))
        assert np.max(np.abs(o2.cpu().numpy() - q_want.reshape(n, 5))) < 3e-5
        assert np.max(np.abs(sg.cpu().numpy() / sg_want.reshape(n, 11) - 1)) < 1e-4


@pytest.mark.parametrize("layer_norm", [False, True])
def test_dropout_training_forward_matches_oracle_stream(ctx, oracle32, layer_norm):
    """A training state's forward draws the step's dropout mask from the library's Philox stream 5; the oracle
    regenerates the same mask from (rate, seed).  Inference is the identity (Keras)."""
    from qbold_vi_amd.ops import TrainState
    B, X, Y, Z = 2, 4, 3, 2
    rate = 0.25
    w, ln, ew = make(ctx, 16, 2, "relu", layer_norm=layer_norm, dropout_rate=rate)
    x = crop_batch(oracle32, B, X, Y, Z, seed=8)
    n = B * X * Y * Z
    st = TrainState(ctx, ew)            # training
    q, ls = st.forward_spatial(dev(x))
    seed = int(ew.shape.dropout_seed)
    assert seed != 0
    q_want, _ = oracle32.encoder_fwd_spatial(w, x, ln=ln, dropout_rate=rate, dropout_seed=seed)
    q_inf, _ = oracle32.encoder_fwd_spatial(w, x, ln=ln)
    assert np.max(np.abs(q.cpu().numpy() - q_want.reshape(n, 5))) < 3e-5
    assert np.max(np.abs(q_want - q_inf)) > 1e-3            # the mask does something
    inf = TrainState(ctx, ew, optimiser_state=False)
    qi, _ = inf.forward_spatial(dev(x))
    assert int(ew.shape.dropout_seed) == 0
    assert np.max(np.abs(qi.cpu().numpy() - q_inf.reshape(n, 5))) < 3e-5
    # another step, another mask
    st.step += 1
    q2, _ = st.forward_spatial(dev(x))
    assert int(ew.shape.dropout_seed) == seed + 1 and not torch.equal(q, q2)
    # the drop rate is what was asked for: the share of zeros among the activations the first normalizer lets through
    from oracle.oracle import Oracle  # noqa: F401  (the mask itself is restated in oracle/qbold_oracle.c: drop_factor)


def _perturbed(w, ln, direction, dln, eps):
    ww = {k: (np.asarray(w[k], np.float64) + eps * direction[k]) if k in direction else w[k] for k in w}
    return ww, (None if ln is None else np.asarray(ln, np.float64) + eps * dln)


@pytest.mark.parametrize("activation,geometry,rate", [("relu", (2, 4, 3, 2), 0.0), ("gelu", (2, 4, 3, 2), 0.2),
                                                      ("relu", (24, 1, 1, 1), 0.3)])
def test_normalizer_weight_gradient_directional(ctx, oracle32, oracle64, activation, geometry, rate):
    """d loss / d (weights, GroupNormalization parameters) along random directions against central differences of the
    float64 oracle, loss = sum of the heads against fixed random cotangents; crops and a voxel batch, with and without
    a dropout mask (the same mask in the backward and in both oracle evaluations)."""
    from qbold_vi_amd.ops import EncoderWeights, TrainState
    B, X, Y, Z = geometry
    U, L = 12, 2
    n = B * X * Y * Z
    w, ln, ew = make(ctx, U, L, activation, dropout_rate=rate, seed=5)
    x = crop_batch(oracle32, B, X, Y, Z, seed=4)
    rng = np.random.default_rng(6)
    g_q = rng.normal(size=(n, 5))
    g_ls = rng.normal(size=(n, 11)) * 0.3
    st = TrainState(ctx, ew)
    q, ls = st.forward_spatial(dev(x))
    seed = int(ew.shape.dropout_seed)
    assert (seed != 0) == (rate > 0)
    grad = st.backward_spatial(dev(g_q.astype(np.float32)), dev(g_ls.astype(np.float32)), None).cpu().numpy().astype(np.float64)
    assert np.isfinite(grad).all()

    def loss(ww, lln):
        with_activation(oracle64, activation)
        try:
            o2, sg = oracle64.encoder_fwd_spatial(ww, x, ln=lln, dropout_rate=rate, dropout_seed=seed)
        finally:
            with_activation(oracle64, "relu")
        return float((o2.reshape(n, 5) * g_q).sum() + (np.log(sg.reshape(n, 11)) * g_ls).sum())

    for trial in range(4):
        direction = {k: rng.standard_normal(np.asarray(w[k]).shape) for k in WEIGHT_NAMES}
        dln = rng.standard_normal(ln.shape)
        if trial == 1:        # GroupNormalization parameters only
            direction = {k: 0 * v for k, v in direction.items()}
        if trial == 2:        # the residual convolutions only
            direction = {k: (v if k in ("Wr1", "Wr2", "br1", "br2") else 0 * v) for k, v in direction.items()}
            dln = 0 * dln
        dflat = EncoderWeights(ctx, 11, U, L, True, -1.0, spatial_taps=9, layer_norm=True).set_from_arrays(
            dict({k: direction[k].astype(np.float32) for k in WEIGHT_NAMES}, ln=dln.astype(np.float32))
        ).flat.cpu().numpy().astype(np.float64)
        eps = 2e-5
        fd = (loss(*_perturbed(w, ln, direction, dln, eps)) - loss(*_perturbed(w, ln, direction, dln, -eps))) / (2 * eps)
        got = float(grad @ dflat)
        assert abs(got - fd) < 1e-2 * (abs(fd) + 0.05), (trial, got, fd)


def test_layer_norm_and_dropout_train_end_to_end(tmp_path, params):
    """use_layer_norm + dropout_rate through the reference-shaped entry: EncoderTrainer builds, a short two-phase run on
    synthetic voxels trains, the weight file carries the GroupNormalization parameters and loads back."""
    import os
    from qbold_vi_amd import training
    from qbold_vi_amd.utils import load_arguments
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = load_arguments(["train.py", os.path.join(root, "configurations", "optimal.yaml")], entry="train")
    args.update(no_units=16, no_intermediate_layers=2, no_pt_epochs=2, no_ft_epochs=3, save_directory=str(tmp_path / "run"),
                synthetic_voxels=4096, use_layer_norm=True, dropout_rate=0.1)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        model, trainer, hist = training.train_model(args, pt_sample_size=200, max_ft_steps=30)
    finally:
        os.chdir(cwd)
    ft = [h for h in hist if "val_elbo" in h]
    assert ft and all(np.isfinite(h["val_elbo"]) for h in ft) and all(np.isfinite(h["loss"]) for h in ft)
    w = np.load(tmp_path / "run" / "final_model.npz")
    assert "ln" in w.files and w["ln"].shape == (2, 4, 16)
    assert np.abs(w["ln"][:, [0, 2]] - 1.0).max() > 1e-4      # gamma moved: the parameters are trained
    model.load_weights(str(tmp_path / "run" / "final_model.npz"))

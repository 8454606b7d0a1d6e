"""GPU tests of the gradient kernels against central finite differences of the float64 oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


@pytest.fixture(scope="module")
def ctx(params):
    """Gradients are checked as exact derivatives of the forward value: node 0's TF-autodiff slope
    is switched off here (its value is pinned by test_signal_bwd_matches_oracle_jacobian)."""
    from qbold_vi_amd.ops import Context
    c = Context(params, full_model=True, include_blood=True)
    c.set_grad_node0(False)
    return c


@pytest.fixture(scope="module")
def oracle64(params):
    """float64 arithmetic, float32-reference semantics of the tissue integral (node 0 removed):
    the function whose derivative the kernels compute, without float32 noise."""
    from oracle.oracle import Oracle
    o = Oracle("f64", params, node0_zero=True)
    yield o
    o.lib.qbo_set_node0_zero(0)   # the policy is process-global in the C library


def kl_stopgrad(o, q_sample, q_logq, prior, zk):
    """mean_k [log q_sg(y_k) - log p(y_k)], y_k = reparam(q_sample, z_k): the q-parameters inside
    log q are stop-gradient in the reference (model.py:596), so finite differences must vary the
    sampling parameters only."""
    K = zk.shape[1]
    acc = 0.0
    for k in range(K):
        y = o.reparam(q_sample, zk[:, k])
        acc = acc + o.logit_mvn_nlogp(y, prior) - o.logit_mvn_nlogp(y, q_logq)
    return acc / K


def _case(oracle32, n, seed):
    from oracle.oracle import init_weights, synth_inputs
    w = init_weights(T=11, U=60, L=2, seed=seed)
    w["gate_offset"] = -3.0
    x, _ = synth_inputs(n, seed=seed, oracle=oracle32)
    prior, q, sigma = oracle32.encoder_fwd(w, x)
    rng = np.random.default_rng(seed)
    q = (q + rng.normal(size=q.shape) * 0.3).astype(np.float32)   # posterior away from the prior
    mask = (rng.uniform(size=n) > 0.25).astype(np.float32)
    return x, mask, q, prior, sigma


@pytest.mark.parametrize("S,K", [(4, 10), (3, 7), (1, 70), (2, 5)])
def test_elbo_head_gradients_vs_oracle_fd(ctx, oracle32, oracle64, S, K):
    """S > 2: a voxel's draws dealt to four lanes (the forward kernels' mapping); S <= 2 -- the reference's training
    defaults are S = 1, K = 70 -- one lane per voxel (elbo_bwd_kernel's LPV).  48 voxels: a partial wave either way."""
    n, seed = 48, 11
    x, mask, q, prior, sigma = _case(oracle32, n, 5)
    ls = np.log(sigma.astype(np.float64))
    zs = oracle32.philox_normals(seed, 0, 0, n, S)
    zk = oracle32.philox_normals(seed, 1, 0, n, K)

    q64 = q.astype(np.float64)

    def loss_v(qq, lss):  # m_v * nll_v + [m_v > 0] * kl_v in float64, table-free literal model
        e = oracle64.elbo(x, mask, qq, prior, np.exp(lss), zs, zk)
        return e["nll_v"] * mask + np.where(mask > 0, kl_stopgrad(oracle64, qq, q64, prior, zk), 0.0)

    sums, gq, gls, nk = ctx.elbo_bwd(dev(x), dev(mask), dev(q), dev(prior), dev(ls.astype(np.float32)),
                                     S, K, seed=seed)
    gq, gls = gq.cpu().numpy(), gls.cpu().numpy()
    # forward values agree with the forward kernel and the oracle
    want = oracle32.elbo(x, mask, q, prior, sigma, zs, zk)
    assert abs((sums[0] + sums[1]).item() / sums[2].item() - want["elbo"]) < 1e-4 * abs(want["elbo"])
    h = 1e-4
    for k in range(5):
        d = np.zeros_like(q64)
        d[:, k] = h
        fd = (loss_v(q64 + d, ls) - loss_v(q64 - d, ls)) / (2 * h)
        scale = np.abs(fd).max() + 1e-3
        assert np.max(np.abs(gq[:, k] - fd)) / scale < 2e-3, (k, np.max(np.abs(gq[:, k] - fd)), scale)
    for t in range(11):
        d = np.zeros_like(ls)
        d[:, t] = h
        fd = (loss_v(q64, ls + d) - loss_v(q64, ls - d)) / (2 * h)
        scale = np.abs(fd).max() + 1e-3
        assert np.max(np.abs(gls[:, t] - fd)) / scale < 2e-3, t


@pytest.mark.parametrize("S", [3, 1])      # 3: four lanes per voxel; 1 (the training default): one lane per voxel
@pytest.mark.parametrize("variant", [dict(student_t_df=2.0, multi_image_normalisation=True),
                                     dict(predict_log_data=True),
                                     dict(student_t_df=5.0, predict_log_data=True, multi_image_normalisation=True)])
def test_elbo_head_gradients_loss_variants(params, oracle64, variant, S):
    """The likelihood switches of EncoderTrainer (model.py:527-568): Student-t (df < 50; the reference's
    sweep configuration uses df = 2), log data, three-image normalisation -- forward value against the
    float32 oracle, head gradients against central differences of the float64 oracle."""
    from oracle.oracle import Oracle
    from qbold_vi_amd.ops import Context
    c = Context(params, True, True, **variant)
    c.set_grad_node0(False)
    o32 = Oracle("f32", params, **variant)
    o64 = Oracle("f64", params, node0_zero=True, **variant)
    n, K, seed = 40, 6, 9
    x, mask, q, prior, sigma = _case(o32, n, 6)
    ls = np.log(sigma.astype(np.float64))
    zs = o32.philox_normals(seed, 0, 0, n, S)
    zk = o32.philox_normals(seed, 1, 0, n, K)
    q64 = q.astype(np.float64)

    def loss_v(qq, lss):
        e = o64.elbo(x, mask, qq, prior, np.exp(lss), zs, zk)
        return e["nll_v"] * mask + np.where(mask > 0, kl_stopgrad(o64, qq, q64, prior, zk), 0.0)

    sums, gq, gls, nk = c.elbo_bwd(dev(x), dev(mask), dev(q), dev(prior), dev(ls.astype(np.float32)), S, K, seed=seed)
    want = o32.elbo(x, mask, q, prior, sigma, zs, zk)
    assert abs((sums[0] + sums[1]).item() / sums[2].item() - want["elbo"]) < 1e-4 * abs(want["elbo"])
    gq, gls = gq.cpu().numpy(), gls.cpu().numpy()
    h = 1e-4
    for k in range(5):
        d = np.zeros_like(q64)
        d[:, k] = h
        fd = (loss_v(q64 + d, ls) - loss_v(q64 - d, ls)) / (2 * h)
        scale = np.abs(fd).max() + 1e-3
        assert np.max(np.abs(gq[:, k] - fd)) / scale < 2e-3, (variant, k)
    for t in (0, 2, 3, 10):
        d = np.zeros_like(ls)
        d[:, t] = h
        fd = (loss_v(q64, ls + d) - loss_v(q64, ls - d)) / (2 * h)
        scale = np.abs(fd).max() + 1e-3
        assert np.max(np.abs(gls[:, t] - fd)) / scale < 2e-3, (variant, t)
    # masked-out voxels carry no gradient
    assert np.abs(gq[mask == 0]).max() == 0.0 and np.abs(gls[mask == 0]).max() == 0.0


def test_elbo_gradient_kl_only_is_exact(ctx, oracle32, oracle64):
    """With the likelihood switched off numerically (huge sigma) the gradient is the KL's, which has
    no Simpson artefact: tight agreement."""
    n, S, K, seed = 32, 2, 40, 3
    x, mask, q, prior, sigma = _case(oracle32, n, 7)
    mask[:] = 1.0
    ls = np.full((n, 11), 12.0)   # sigma = e^12: residual term vanishes
    zs = oracle32.philox_normals(seed, 0, 0, n, S)
    zk = oracle32.philox_normals(seed, 1, 0, n, K)

    q64 = q.astype(np.float64)

    def loss_v(qq):
        return kl_stopgrad(oracle64, qq, q64, prior, zk)

    _, gq, _, _ = ctx.elbo_bwd(dev(x), dev(mask), dev(q), dev(prior), dev(ls.astype(np.float32)), S, K,
                               seed=seed)
    gq = gq.cpu().numpy()
    for k in range(5):
        d = np.zeros_like(q64)
        d[:, k] = 1e-5
        fd = (loss_v(q64 + d) - loss_v(q64 - d)) / 2e-5
        assert np.max(np.abs(gq[:, k] - fd)) < 2e-4 * (np.abs(fd).max() + 1.0), k


def _weights(ctx, U=60, L=2, cw=True, seed=3):
    from oracle.oracle import init_weights
    from qbold_vi_amd.ops import EncoderWeights
    w = init_weights(T=11, U=U, L=L, channelwise_gating=cw, seed=seed)
    rng = np.random.default_rng(seed)
    for k in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
    w["gate_offset"] = -1.0   # keep the residual branch well alive for the gradient check
    ew = EncoderWeights(ctx, 11, U, L, cw, -1.0).set_from_arrays(w)
    return w, ew


def _perturbed(w, direction, eps):
    out = dict(w)
    for k, d in direction.items():
        out[k] = (w[k].astype(np.float64) + eps * d)
    return out


@pytest.mark.parametrize("U,L,cw", [(60, 2, True), (24, 1, False)])
def test_training_forward_matches_inference_and_oracle(ctx, oracle32, U, L, cw):
    from oracle.oracle import synth_inputs
    from qbold_vi_amd.ops import TrainState
    w, ew = _weights(ctx, U, L, cw)
    x, _ = synth_inputs(777, seed=2, oracle=oracle32)
    st = TrainState(ctx, ew)
    q2, ls = st.forward(dev(x), 2)
    w1, w2, wsg = oracle32.encoder_fwd(w, x)
    assert np.abs(q2.cpu().numpy() - w2).max() < 2e-5
    assert np.abs(ls.cpu().numpy() - np.log(wsg)).max() < 2e-5
    q1, none = st.forward(dev(x), 1)
    assert none is None and np.abs(q1.cpu().numpy() - w1).max() < 2e-5
    o1, o2, sg = ctx.encoder_fwd(ew, dev(x))
    assert torch.allclose(o2, q2, atol=2e-5) and torch.allclose(o1, q1, atol=2e-5)


def test_inverse_gamma_prior_loss_and_head_gradient(ctx, oracle32, oracle64):
    """synthetic_data_loss with inv_gamma_alpha * inv_gamma_beta > 0 (model.py:492-507; the sweep
    configuration of the reference uses it): value against the oracle, head gradient against central
    differences of the float64 oracle."""
    from qbold_vi_amd.ops import EncoderWeights, TrainState
    rng = np.random.default_rng(21)
    n, a, b = 400, 3.0, 0.15
    q = (rng.normal(size=(n, 5)) * 0.5).astype(np.float32)
    y3 = np.stack([rng.uniform(0.1, 0.7, n), rng.uniform(0.01, 0.15, n), np.ones(n)], -1).astype(np.float32)
    st = TrainState(ctx, EncoderWeights(ctx, 11, 60, 2, True, -3.0), optimiser_state=False)
    lv, gq = st.synth_loss_bwd(dev(y3), dev(q), a, b)
    want = oracle32.synthetic_data_loss(y3, q, a, b)
    assert abs(float(lv.double().mean()) - want) < 1e-5 * abs(want)
    lv0, gq0 = st.synth_loss_bwd(dev(y3), dev(q))
    assert abs(float(lv0.double().mean()) - oracle32.synthetic_data_loss(y3, q)) < 1e-5 * abs(want)
    assert float((lv - lv0).abs().max()) > 1e-3      # the prior term is there
    g = gq.cpu().numpy().astype(np.float64) * n       # per-voxel d loss_v / d q
    q64 = q.astype(np.float64)
    eps = 1e-5
    for k in range(5):
        d = np.zeros((n, 5)); d[:, k] = eps
        # per-voxel losses are independent: N * d(mean)/dq_v,k via one-voxel-at-a-time is the same as
        # perturbing all voxels at once and differencing the per-voxel terms
        lp = np.array([oracle64.synthetic_data_loss(y3[i:i + 1], q64[i:i + 1] + d[i:i + 1], a, b) for i in range(0, n, 8)])
        lm = np.array([oracle64.synthetic_data_loss(y3[i:i + 1], q64[i:i + 1] - d[i:i + 1], a, b) for i in range(0, n, 8)])
        fd = (lp - lm) / (2 * eps)
        got = g[::8, k]
        assert np.max(np.abs(got - fd) / (np.abs(fd) + 1e-2)) < 2e-3, k
    # API level: the trainer's loss takes the same arguments as the reference's
    from qbold_vi_amd import EncoderTrainer
    import configparser, os
    cp = configparser.ConfigParser()
    cp.read(os.path.join(os.path.dirname(__file__), "..", "config"))
    tr = EncoderTrainer(system_params=dict(cp["DEFAULT"]), no_units=60, use_layer_norm=False, dropout_rate=0.0,
                        no_intermediate_layers=2, student_t_df=200, initial_im_sigma=0.05, activation_type='relu',
                        multi_image_normalisation=False, channelwise_gating=True, infer_inv_gamma=False,
                        use_population_prior=False, use_mvg=True, predict_log_data=False)
    got = float(tr.synthetic_data_loss(dev(y3).reshape(n, 1, 1, 1, 3), dev(q).reshape(n, 1, 1, 1, 5), False, a, b))
    assert abs(got - want) < 1e-5 * abs(want)
    # use_r2p_loss adds the R2' term (its own test below): the value changes and stays finite
    with_r2p = float(tr.synthetic_data_loss(dev(y3).reshape(n, 1, 1, 1, 3), dev(q).reshape(n, 1, 1, 1, 5), True, a, b))
    assert np.isfinite(with_r2p) and abs(with_r2p - got) > 1e-3


@pytest.mark.parametrize("U,L,cw", [(60, 2, True), (24, 1, False), (128, 1, True), (80, 2, False), (96, 1, True)])
def test_pretraining_weight_gradient_directional(ctx, oracle32, oracle64, U, L, cw):
    """d/dw mean_v -log p(y_v; q1(x_v; w)) against central differences of the float64 oracle along
    random directions in weight space."""
    from oracle.oracle import WEIGHT_NAMES, synth_inputs
    from qbold_vi_amd.ops import TrainState
    w, ew = _weights(ctx, U, L, cw)
    n = 512
    x, y = synth_inputs(n, seed=4, oracle=oracle32)
    y3 = np.concatenate([y, y[:, :1]], -1).astype(np.float32)
    st = TrainState(ctx, ew)
    q1, _ = st.forward(dev(x), 1)
    lv, gq = st.synth_loss_bwd(dev(y3), q1)
    grad = st.backward(1, gq).cpu().numpy().astype(np.float64)
    want_loss = oracle32.synthetic_data_loss(y3, oracle32.encoder_fwd(w, x)[0])
    assert abs(float(lv.mean()) - want_loss) < 1e-4 * abs(want_loss) + 1e-5

    def loss(ww):
        return oracle64.synthetic_data_loss(y3, oracle64.encoder_fwd(ww, x)[0])

    from qbold_vi_amd.ops import EncoderWeights
    rng = np.random.default_rng(0)
    for trial in range(4):
        direction = {k: rng.standard_normal(w[k].shape) for k in WEIGHT_NAMES}
        if trial == 1:   # only the stream-1 tensors
            for k in ("Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Ws", "bs"):
                direction[k] *= 0
        dflat = EncoderWeights(ctx, 11, U, L, cw, -1.0).set_from_arrays(
            {k: direction[k].astype(np.float32) for k in WEIGHT_NAMES}).flat.cpu().numpy().astype(np.float64)
        # float64 oracle: a small step keeps the number of relu kinks crossed (an O(eps) error in the
        # difference quotient each) negligible; the tolerance is relative to the size of the per-tensor
        # contributions, which can cancel in the total
        eps = 2e-6
        fd = (loss(_perturbed(w, direction, eps)) - loss(_perturbed(w, direction, -eps))) / (2 * eps)
        got = float(grad @ dflat)
        scale = abs(fd) + 1e-3 * float(np.abs(grad * dflat).sum())
        assert abs(got - fd) < 5e-3 * (scale + 1e-2), (trial, got, fd, scale)
    # tensors that stream 1 does not touch get exactly zero gradient
    g = EncoderWeights(ctx, 11, U, L, cw, -1.0)
    g.flat.copy_(torch.as_tensor(grad, dtype=torch.float32))
    ga = g.to_arrays()
    for k in ("Wr1", "Wr2", "Wg", "Ws", "bs"):
        assert np.all(ga[k] == 0)


@pytest.mark.parametrize("U,L,cw", [(60, 2, True), (24, 1, False), (128, 1, True), (80, 2, False)])
def test_finetune_weight_gradient_directional(ctx, oracle32, oracle64, U, L, cw):
    """d/dw of the masked-mean negative ELBO through encoder stream 2 + sampling, along random
    directions, against the float64 oracle (stop-gradient KL emulated as in kl_stopgrad)."""
    from oracle.oracle import WEIGHT_NAMES, synth_inputs
    from qbold_vi_amd.ops import EncoderWeights, TrainState
    w, ew = _weights(ctx, U, L, cw)
    n, S, K, seed = 256, 2, 6, 21
    x, _ = synth_inputs(n, seed=6, oracle=oracle32)
    rng = np.random.default_rng(1)
    mask = (rng.uniform(size=n) > 0.2).astype(np.float32)
    prior = oracle32.encoder_fwd(w, x)[0]
    st = TrainState(ctx, ew)
    q2, ls = st.forward(dev(x), 2)
    sums, gq, gls, _ = ctx.elbo_bwd(dev(x), dev(mask), q2, dev(prior), ls, S, K, seed=seed)
    grad = st.backward(2, gq, gls, sums).cpu().numpy().astype(np.float64)
    zs = oracle32.philox_normals(seed, 0, 0, n, S)
    zk = oracle32.philox_normals(seed, 1, 0, n, K)
    q_fixed = oracle64.encoder_fwd(w, x)[1]

    def loss(ww):
        _, qq, sg = oracle64.encoder_fwd(ww, x)
        e = oracle64.elbo(x, mask, qq, prior, sg, zs, zk)
        kl = kl_stopgrad(oracle64, qq, q_fixed, prior, zk)
        return ((e["nll_v"] * mask).sum() + np.where(mask > 0, kl, 0).sum()) / mask.sum()

    for trial in range(4):
        direction = {k: rng.standard_normal(w[k].shape) for k in WEIGHT_NAMES}
        if trial == 1:   # sigma head only
            for k in WEIGHT_NAMES:
                if k not in ("Ws", "bs"):
                    direction[k] *= 0
        if trial == 2:   # residual branch only
            for k in WEIGHT_NAMES:
                if k not in ("Wr1", "br1", "Wr2", "br2", "Wg", "bg"):
                    direction[k] *= 0
        dflat = EncoderWeights(ctx, 11, U, L, cw, -1.0).set_from_arrays(
            {k: direction[k].astype(np.float32) for k in WEIGHT_NAMES}).flat.cpu().numpy().astype(np.float64)
        eps = 2e-6
        fd = (loss(_perturbed(w, direction, eps)) - loss(_perturbed(w, direction, -eps))) / (2 * eps)
        got = float(grad @ dflat)
        assert abs(got - fd) < 1e-2 * (abs(fd) + 0.05), (trial, got, fd)


def test_block_backward_kernel_against_the_float64_oracle(ctx, oracle32, oracle64):
    """block_bwd_dw_kernel -- one gated block's whole backward, its four weight gradients accumulated in registers --
    DIRECTLY against central differences of the float64 oracle at the training shape (U = 60, L = 2, 4,096 voxels,
    S = 1, K = 70): one direction per block matrix (Wc, Wr1, Wr2, Wg of either block) and their biases, so that each
    of the kernel's accumulators is held to the oracle on its own, not only to the layer-wise HIP path."""
    import ctypes as C
    from oracle.oracle import WEIGHT_NAMES, synth_inputs
    from qbold_vi_amd.ops import EncoderWeights, TrainState
    U, L = 60, 2
    w, ew = _weights(ctx, U, L, True)
    n, S, K, seed = 4096, 1, 70, 33
    assert int(ctx.lib.qbold_encoder_train_bwd_recomputes(ctx.handle, C.byref(ew.shape), n)) == 2   # the kernel under test runs
    x, _ = synth_inputs(n, seed=8, oracle=oracle32)
    rng = np.random.default_rng(2)
    mask = (rng.uniform(size=n) > 0.1).astype(np.float32)
    prior = oracle32.encoder_fwd(w, x)[0]
    st = TrainState(ctx, ew)
    q2, ls = st.forward(dev(x), 2)
    sums, gq, gls, _ = ctx.elbo_bwd(dev(x), dev(mask), q2, dev(prior), ls, S, K, seed=seed)
    grad = st.backward(2, gq, gls, sums).cpu().numpy().astype(np.float64)
    zs = oracle32.philox_normals(seed, 0, 0, n, S)
    zk = oracle32.philox_normals(seed, 1, 0, n, K)
    q_fixed = oracle64.encoder_fwd(w, x)[1]

    def loss(ww):
        _, qq, sg = oracle64.encoder_fwd(ww, x)
        e = oracle64.elbo(x, mask, qq, prior, sg, zs, zk)
        kl = kl_stopgrad(oracle64, qq, q_fixed, prior, zk)
        return ((e["nll_v"] * mask).sum() + np.where(mask > 0, kl, 0).sum()) / mask.sum()

    for name in ("Wc", "Wr1", "Wr2", "Wg", "bc", "br1", "br2", "bg"):
        for blk in range(L):
            direction = {k: np.zeros(w[k].shape) for k in WEIGHT_NAMES}
            direction[name][blk] = rng.standard_normal(w[name][blk].shape)
            dflat = EncoderWeights(ctx, 11, U, L, True, -1.0).set_from_arrays(
                {k: direction[k].astype(np.float32) for k in WEIGHT_NAMES}).flat.cpu().numpy().astype(np.float64)
            eps = 2e-6
            fd = (loss(_perturbed(w, direction, eps)) - loss(_perturbed(w, direction, -eps))) / (2 * eps)
            got = float(grad @ dflat)
            assert abs(got - fd) < 5e-3 * (abs(fd) + 0.02), (name, blk, got, fd)


def test_adamw_matches_numpy(ctx):
    from qbold_vi_amd.ops import TrainState
    _, ew = _weights(ctx, 24, 1, True)
    st = TrainState(ctx, ew)
    rng = np.random.default_rng(0)
    w = ew.flat.cpu().numpy().astype(np.float64)
    m = np.zeros_like(w)
    v = np.zeros_like(w)
    for t in range(1, 4):
        g = rng.standard_normal(w.shape)
        st.grad.copy_(torch.as_tensor(g, dtype=torch.float32))
        lr, wd, b1, b2, eps = 5e-3 / t, 2e-4, 0.9, 0.9, 1e-7
        st.adamw(lr, wd, b1, b2, eps)
        w = w - wd * w                      # tfa DecoupledWeightDecay: before the Adam update
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        w = w - lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)
        np.testing.assert_allclose(ew.flat.cpu().numpy(), w, rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
def test_r2p_loss_term_value_and_gradient(ctx, oracle32, oracle64):
    """use_r2p_loss (model.py:475-490): ten reparameterised draws, r = dw(OEF) DBV, a normal fitted by the
    draws' mean and biased std, gaussian_nll (:403-404) of the true R2'.  Value against the oracle's
    posterior moments (float32 and float64) on the same explicit normals; head gradient against central
    differences of the float64 expression; the in-kernel Philox stream against the oracle's; accumulation
    into caller buffers."""
    rng = np.random.default_rng(33)
    n, ns = 300, 10
    q = (rng.normal(size=(n, 5)) * 0.5).astype(np.float32)
    z = rng.normal(size=(n, ns, 2)).astype(np.float32)
    y3 = np.stack([rng.uniform(0.1, 0.7, n), rng.uniform(0.01, 0.15, n), rng.uniform(1.0, 8.0, n)], -1).astype(np.float32)

    def nll_ref(orc, qq):
        m, v = orc.moments(qq, z.astype(orc.dtype))
        m, v = m[:, 2].astype(np.float64), v[:, 2].astype(np.float64)
        return 0.5 * np.log(v) + 0.5 * (y3[:, 2] - m) ** 2 / v

    lv, gq = ctx.r2p_loss_bwd(dev(y3), dev(q), ns, z=dev(z))
    want64 = nll_ref(oracle64, q.astype(np.float64))
    got = lv.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - want64) / (np.abs(want64) + 1.0)) < 2e-4
    assert np.max(np.abs(got - nll_ref(oracle32, q)) / (np.abs(want64) + 1.0)) < 2e-4
    g = gq.cpu().numpy().astype(np.float64)
    eps = 1e-5
    for k in range(5):
        d = np.zeros((n, 5)); d[:, k] = eps
        fd = (nll_ref(oracle64, q.astype(np.float64) + d) - nll_ref(oracle64, q.astype(np.float64) - d)) / (2 * eps)
        scale = np.abs(fd) + 0.05 * np.abs(g).max(axis=1) + 1e-2
        assert np.max(np.abs(g[:, k] - fd) / scale) < 5e-3, k
    # accumulation and scale: adds scale * gradient and the value to the caller's buffers
    lv2 = torch.full((n,), 2.0, device="cuda"); gq2 = torch.ones((n, 5), device="cuda")
    ctx.r2p_loss_bwd(dev(y3), dev(q), ns, z=dev(z), scale=0.25, g_q=gq2, loss_v=lv2)
    assert torch.allclose(lv2, lv + 2.0, rtol=1e-6, atol=1e-6)
    assert torch.allclose(gq2, 1.0 + 0.25 * gq, rtol=1e-5, atol=1e-6)
    # in-kernel Philox stream 3 = the oracle's normals for (seed, voxel0)
    zp = oracle32.philox_normals(5, 3, 1000, n, ns)
    lv3, gq3 = ctx.r2p_loss_bwd(dev(y3), dev(q), ns, seed=5, voxel0=1000)
    lv4, gq4 = ctx.r2p_loss_bwd(dev(y3), dev(q), ns, z=dev(zp))
    assert torch.allclose(lv3, lv4, rtol=2e-4, atol=2e-4)
    assert float((gq3 - gq4).abs().max()) < 2e-3 * float(gq4.abs().max())
    # odd draw counts use half a Philox call; fewer than two draws have no spread
    lv5, _ = ctx.r2p_loss_bwd(dev(y3), dev(q), 7, z=dev(z[:, :7].copy()))
    m7, v7 = oracle64.moments(q.astype(np.float64), z[:, :7].astype(np.float64))
    w7 = 0.5 * np.log(v7[:, 2]) + 0.5 * (y3[:, 2] - m7[:, 2]) ** 2 / v7[:, 2]
    assert np.max(np.abs(lv5.cpu().numpy() - w7) / (np.abs(w7) + 1.0)) < 2e-4
    with pytest.raises(Exception):
        ctx.r2p_loss_bwd(dev(y3), dev(q), 1)


@pytest.mark.gpu
def test_weight_gradients_add_over_ragged_batches(ctx, oracle32):
    """Batch sizes that are no multiple of the kernels' 4-voxel MFMA steps or 16-voxel tiles: the weight
    gradient of a batch is the size-weighted sum of its parts' gradients (both streams' GEMM, weight-gradient
    and slab-reduction kernels with ragged tails; one voxel alone as the smallest case)."""
    from oracle.oracle import synth_inputs
    from qbold_vi_amd.ops import TrainState
    w, ew = _weights(ctx, 60, 2, True)
    n = 1003
    x, y = synth_inputs(n, seed=8, oracle=oracle32)
    y3 = dev(np.concatenate([y, y[:, :1]], -1).astype(np.float32))
    xd = dev(x)
    st = TrainState(ctx, ew)

    def grad_of(lo, hi):
        q1, _ = st.forward(xd[lo:hi].contiguous(), 1)
        _, gq = st.synth_loss_bwd(y3[lo:hi].contiguous(), q1)      # already divided by hi - lo
        return st.backward(1, gq).double().clone() * (hi - lo)

    full = grad_of(0, n)
    parts = grad_of(0, 333) + grad_of(333, 1002) + grad_of(1002, 1003)
    scale = float(full.abs().max())
    assert float((full - parts).abs().max()) < 2e-5 * scale
    # stream 2 through the ELBO head gradients
    mask = torch.ones(n, device="cuda")
    prior = ctx.encoder_fwd(ew, xd, want=("out1",))[0]

    def grad2_of(lo, hi):
        xs = xd[lo:hi].contiguous()
        q, ls = st.forward(xs, 2)
        sums, gq, gls, _ = ctx.elbo_bwd(xs, mask[lo:hi].contiguous(), q, prior[lo:hi].contiguous(), ls, 1, 6,
                                        seed=3, voxel0=lo)
        return st.backward(2, gq, gls, sums).double().clone() * (hi - lo)

    full2 = grad2_of(0, n)
    parts2 = grad2_of(0, 333) + grad2_of(333, 1002) + grad2_of(1002, 1003)
    assert float((full2 - parts2).abs().max()) < 5e-5 * float(full2.abs().max())


def test_learned_inverse_gamma_hyper_prior(params):
    """infer_inv_gamma with the diagonal family (model.py:201-205, 454-455, 493-507): four exp-activated scalars
    (alpha, beta of the OEF and DBV variance priors) appended to the first output; synthetic_data_loss subtracts
    log IG(exp(2 s_o); a_o, b_o) + log IG(exp(2 s_d); a_d, b_d).  Value against scipy.stats.invgamma (= tfp
    InverseGamma), head gradient and hyper-gradient against central differences of that float64 restatement."""
    from scipy.stats import invgamma
    from qbold_vi_amd import EncoderTrainer
    from qbold_vi_amd.training import HyperPriorState
    tr = EncoderTrainer(params, no_intermediate_layers=1, no_units=16, activation_type='relu', use_mvg=False,
                        infer_inv_gamma=True, use_population_prior=False, channelwise_gating=True,
                        multi_image_normalisation=False, predict_log_data=False)
    model, _ = tr.create_encoder(gate_offset=-3.0, resid_init_std=0.05, no_ip_images=11)
    np.testing.assert_allclose(model.hyper_params(), [20.0, 2.5, 20.0, 2.5])
    rng = np.random.default_rng(3)
    n = 777
    x = torch.as_tensor(rng.uniform(0.3, 1.0, (n, 1, 1, 1, 11)).astype(np.float32), device="cuda")
    out1 = model.predict(x, want=("out1",))[0]
    assert out1.shape[-1] == 8 and torch.allclose(out1[..., 4:], torch.tensor([20.0, 2.5, 20.0, 2.5], device="cuda"))
    q4 = out1.reshape(n, 8)[:, :4].double().cpu().numpy() + rng.normal(size=(n, 4)) * 0.3
    y = np.stack([rng.uniform(0.1, 0.7, n), rng.uniform(0.01, 0.15, n), rng.uniform(1, 10, n)], -1).astype(np.float32)
    hyper = np.array([7.0, 0.8, 11.0, 1.7])
    model.hyper_raw = np.log(hyper)

    def ref_loss(q, h):   # float64: diagonal logit-normal NLL (model.py:406-421) minus the inverse-gamma log-densities
        s_o, s_d = 3 * np.tanh(q[:, 1]) - 1, 3 * np.tanh(q[:, 3]) - 1
        xo, xd = (y[:, 0].astype(np.float64) - 0.04) / 0.8, (y[:, 1].astype(np.float64) - 0.001) / 0.2
        lo, ld = np.log(xo / (1 - xo)), np.log(xd / (1 - xd))
        nll = s_o + 0.5 * ((lo - q[:, 0]) / np.exp(s_o)) ** 2 + s_d + 0.5 * ((ld - q[:, 2]) / np.exp(s_d)) ** 2 \
            + np.log(xo * (1 - xo)) + np.log(xd * (1 - xd))
        prior = invgamma.logpdf(np.exp(2 * s_o), a=h[0], scale=h[1]) + invgamma.logpdf(np.exp(2 * s_d), a=h[2], scale=h[3])
        return (nll - prior).mean()

    q8 = torch.cat([torch.as_tensor(q4, dtype=torch.float32, device="cuda"),
                    torch.as_tensor(hyper, dtype=torch.float32, device="cuda").expand(n, 4)], -1)
    got = float(tr.synthetic_data_loss(torch.as_tensor(y, device="cuda"), q8.reshape(n, 1, 1, 1, 8)))
    want = ref_loss(q4, hyper)
    assert abs(got - want) < 2e-5 * max(1.0, abs(want))
    # gradients: head (through qbold_synth_loss_bwd + qbold_hyper_prior_bwd) and the four hyper-parameters
    from qbold_vi_amd.ops import TrainState
    st = TrainState(tr.context, model.weights)
    q5 = torch.cat([q8[:, :4], torch.zeros(n, 1, device="cuda")], -1).contiguous()
    lv, gq = st.synth_loss_bwd(torch.as_tensor(y, device="cuda"), q5)
    stats = tr.context.hyper_prior_bwd(q5, hyper, scale=1.0 / n, g_q=gq, loss_v=lv)
    assert abs(float(lv.mean()) - 1.8378770664093453 - want) < 2e-5 * max(1.0, abs(want))
    g = gq.cpu().numpy().astype(np.float64)
    eps = 1e-5
    for v, k in ((0, 1), (5, 3), (100, 0), (333, 1), (776, 3)):
        qp, qm = q4.copy(), q4.copy()
        qp[v, k] += eps
        qm[v, k] -= eps
        fd = (ref_loss(qp, hyper) - ref_loss(qm, hyper)) / (2 * eps)
        assert abs(g[v, k] - fd) < 2e-3 * abs(fd) + 1e-7, (v, k, g[v, k], fd)
    hs = HyperPriorState(model)
    gh = hs.gradient(stats.cpu().numpy(), n)          # d loss / d log(hyper)
    for k in range(4):
        hp, hm = hyper.copy(), hyper.copy()
        hp[k] *= np.exp(eps)
        hm[k] *= np.exp(-eps)
        fd = (ref_loss(q4, hp) - ref_loss(q4, hm)) / (2 * eps)
        assert abs(gh[k] - fd) < 1e-4 * abs(fd) + 1e-6, (k, gh[k], fd)
    # the reference cannot run the pair (infer_inv_gamma, use_mvg=True): model.py:455 splits 9 channels in two
    with pytest.raises(NotImplementedError):
        EncoderTrainer(params, activation_type='relu', use_mvg=True, infer_inv_gamma=True, use_population_prior=False)


@pytest.mark.parametrize("T,L,taps", [(11, 2, 1), (11, 1, 1), (24, 2, 1), (11, 2, 9)])
def test_one_launch_training_forward_and_block_backward(params, T, L, taps, monkeypatch):
    """Voxel batches of the LDS-resident shapes train through two fused kernels: qbold_encoder_train_fwd_fused
    (stream 2 forward, every saved tensor written once) and block_bwd_kernel (a gated block's data-side backward
    in one launch: block recomputed, deltas scaled per voxel into the f16 split's range).  Against the layer-wise
    exact-f32 kernels: heads, every saved tensor the backward reads (n: columns < T; h and each block's output:
    columns < U), and the weight gradient -- with head gradients of the size a real loss
    produces (divided by the voxel count: 1e-6 and below, far under f16's normal range).  Ragged N."""
    from qbold_vi_amd.init import init_encoder_weights
    from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
    p = dict(params)
    if T == 24:
        p.update(tau_start="-0.028", tau_end="0.065", tau_step="0.004")
    U, N = 60, 16 * 37 + 5
    w = init_encoder_weights(T=T, U=U, L=L, channelwise_gating=True, resid_init_std=0.3, im_loss_sigma=0.05, seed=4,
                             spatial_taps=taps)      # 9: Keras 3x3x1 kernels, of which a voxel batch sees the centre tap
    rng = np.random.default_rng(11)
    x = torch.as_tensor(rng.uniform(0.2, 1.0, (N, T)).astype(np.float32), device="cuda")
    scale = np.exp(rng.uniform(np.log(1e-9), np.log(1e-4), (N, 1)))      # per-voxel magnitudes over five decades
    g_q = torch.as_tensor((rng.normal(size=(N, 5)) * scale).astype(np.float32), device="cuda")
    g_ls = torch.as_tensor((rng.normal(size=(N, T)) * scale).astype(np.float32), device="cuda")
    out = {}
    for fused in (False, True):
        ctx = Context(p, True, True)
        # QBOLD_KSEL_LAYERWISE_BWD = 131072: the layer-wise backward (gate_bwd_kernel + xw64 launches);
        # QBOLD_KSEL_HEADS_BWD_LAYERWISE = 1048576: the heads' backward through a delta tensor
        ctx.set_kernel_selection(0 if fused else 131072 | 1048576)
        ew = EncoderWeights(ctx, T, U, L, True, -3.0, spatial_taps=taps).set_from_arrays(w)
        st = TrainState(ctx, ew)
        st.fused_forward = fused
        st.workspace(N).fill_(float("nan"))      # whatever the backward reads must have been written
        q, ls = st.forward(x, 2)
        slots = st.workspace(N)[: (2 + 5 * L) * N * 64].reshape(2 + 5 * L, N, 64).clone()
        grad = st.backward(2, g_q, g_ls).clone()
        out[fused] = (q.clone(), ls.clone(), slots, grad)
    (q0, ls0, s0, g0), (q1, ls1, s1, g1) = out[False], out[True]
    assert torch.isfinite(q1).all() and torch.isfinite(ls1).all() and torch.isfinite(g1).all()
    assert (q1 - q0).abs().max() < 1e-5 * max(1.0, float(q0.abs().max()))      # split-f16 products vs exact f32
    assert (ls1 - ls0).abs().max() < 1e-5 * max(1.0, float(ls0.abs().max()))
    assert (s1[0, :, :T] - s0[0, :, :T]).abs().max() < 2e-6           # n = log(x / x_se)
    assert (s1[0, :, T:(T + 3) & ~3] == 0).all()
    for k in range(1, 2 + 5 * L):
        if k >= 2 and (k - 2) % 5 != 4:
            continue      # skip, t, r and the gate logits: recomputed by the block backward, left out by the forward
        d = (s1[k, :, :U] - s0[k, :, :U]).abs().max()
        assert d < 3e-5 * max(1.0, float(s0[k, :, :U].abs().max())), (k, float(d))
    # weight gradient, tensor by tensor (relative to each tensor's own largest entry)
    for name, pieces in ew._slices().items():
        for l, (off, shape) in enumerate(pieces):
            cnt = int(np.prod(shape))
            a, b = g0[off:off + cnt], g1[off:off + cnt]
            assert (a - b).abs().max() <= 2e-4 * float(a.abs().max()) + 1e-30, \
                (name, l, float((a - b).abs().max()), float(a.abs().max()))


def test_fused_training_kernels_over_widths_and_batch_sizes(params, monkeypatch):
    """The fused training kernels against the layer-wise exact-f32 path over widths that pad the 64-unit tiles
    differently (8 .. 64), one and two blocks, and batch sizes from a single voxel to several workgroups' worth."""
    from qbold_vi_amd.init import init_encoder_weights
    from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
    rng = np.random.default_rng(2024)
    cases = [(8, 1, 1), (16, 2, 17), (33, 2, 64), (48, 1, 255), (49, 2, 256), (60, 2, 4097), (64, 2, 1000), (64, 1, 16)]
    for U, L, N in cases:
        w = init_encoder_weights(T=11, U=U, L=L, channelwise_gating=True, resid_init_std=0.3, im_loss_sigma=0.05,
                                 seed=U + L)
        x = torch.as_tensor(rng.uniform(0.2, 1.0, (N, 11)).astype(np.float32), device="cuda")
        g_q = torch.as_tensor((rng.normal(size=(N, 5)) / N).astype(np.float32), device="cuda")
        g_ls = torch.as_tensor((rng.normal(size=(N, 11)) / N).astype(np.float32), device="cuda")
        out = {}
        for fused in (False, True):
            ctx = Context(params, True, True)
            ctx.set_kernel_selection(0 if fused else 131072 | 1048576)   # QBOLD_KSEL_LAYERWISE_BWD | _HEADS_BWD_LAYERWISE
            ew = EncoderWeights(ctx, 11, U, L, True, -3.0).set_from_arrays(w)
            st = TrainState(ctx, ew)
            st.fused_forward = fused
            st.workspace(N).fill_(float("nan"))
            q, ls = st.forward(x, 2)
            out[fused] = (q.clone(), ls.clone(), st.backward(2, g_q, g_ls).clone())
        (q0, ls0, g0), (q1, ls1, g1) = out[False], out[True]
        assert torch.isfinite(g1).all(), (U, L, N)
        assert (q1 - q0).abs().max() < 1e-5 * max(1.0, float(q0.abs().max())), (U, L, N)
        assert (ls1 - ls0).abs().max() < 1e-5 * max(1.0, float(ls0.abs().max())), (U, L, N)
        for name, pieces in ew._slices().items():
            for l, (off, shape) in enumerate(pieces):
                cnt = int(np.prod(shape))
                a, b = g0[off:off + cnt], g1[off:off + cnt]
                assert (a - b).abs().max() <= 3e-4 * float(a.abs().max()) + 1e-30, \
                    (U, L, N, name, l, float((a - b).abs().max()), float(a.abs().max()))



def test_voxel_batch_weight_gradients_bf16_pieces_and_one_pass_heads_against_exact_f32(params):
    """The layer-wise backward of a voxel batch with its 1 x 1 weight gradients on the bf16 matrix pipe (xtdb_kernel,
    three bfloat16 pieces per float32 operand) and both heads in one pass (heads_bwd_kernel), against the same
    backward with QBOLD_KSEL_DW_EXACT_F32 | QBOLD_KSEL_HEADS_BWD_LAYERWISE (exact float32 products, delta tensor):
    ragged batches from a single voxel to several workgroups' worth, widths that pad the 64-unit rows differently
    (widths that are no multiple of 4 take the exact kernels either way), deltas over five decades."""
    from qbold_vi_amd.init import init_encoder_weights
    from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
    rng = np.random.default_rng(77)
    layerwise = 131072                      # QBOLD_KSEL_LAYERWISE_BWD: every weight gradient through xtd / xtdb
    for U, L, N in [(8, 1, 1), (20, 2, 17), (33, 1, 64), (60, 2, 255), (60, 2, 4097), (64, 2, 70001), (64, 1, 16)]:
        w = init_encoder_weights(T=11, U=U, L=L, channelwise_gating=True, resid_init_std=0.3, im_loss_sigma=0.05,
                                 seed=U + L)
        x = torch.as_tensor(rng.uniform(0.2, 1.0, (N, 11)).astype(np.float32), device="cuda")
        scale = np.exp(rng.uniform(np.log(1e-9), np.log(1e-4), (N, 1)))
        g_q = torch.as_tensor((rng.normal(size=(N, 5)) * scale).astype(np.float32), device="cuda")
        g_ls = torch.as_tensor((rng.normal(size=(N, 11)) * scale).astype(np.float32), device="cuda")
        grads = {}
        for sel in (layerwise, layerwise | 4194304, layerwise | 524288 | 1048576):
            ctx = Context(params, True, True)
            ctx.set_kernel_selection(sel)
            ew = EncoderWeights(ctx, 11, U, L, True, -3.0).set_from_arrays(w)
            st = TrainState(ctx, ew)
            st.fused_forward = False
            st.workspace(N).fill_(float("nan"))
            st.forward(x, 2)
            grads[sel] = st.backward(2, g_q, g_ls).double().clone()
        a = grads[layerwise | 524288 | 1048576]
        for sel in (layerwise, layerwise | 4194304):   # two f16 halves under the wave's delta scale; three bf16 pieces
            b = grads[sel]
            assert torch.isfinite(b).all(), (U, L, N, sel)
            for name, pieces in ew._slices().items():
                for l, (off, shape) in enumerate(pieces):
                    cnt = int(np.prod(shape))
                    ta, tb = a[off:off + cnt], b[off:off + cnt]
                    assert float((ta - tb).abs().max()) <= 3e-6 * float(ta.abs().max()) + 1e-30, \
                        (U, L, N, sel, name, l, float((ta - tb).abs().max()), float(ta.abs().max()))


def test_crop_weight_gradients_queued_slab_sums_equal_the_separate_launches(params):
    """The crop backward queues its weight-gradient slab sums into a few launches (slab_reduce_jobs_kernel);
    QBOLD_KSEL_SLAB_SUMS_SEPARATE sums every group of slabs with a launch of its own.  Both add the same slabs in the
    same order: the gradients are equal bit for bit, for crops small enough to leave most slabs empty and for
    batches that fill the slab region more than once."""
    from qbold_vi_amd.init import init_encoder_weights
    from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
    rng = np.random.default_rng(91)
    separate = 2097152                      # QBOLD_KSEL_SLAB_SUMS_SEPARATE
    for U, L, geometry, act in [(12, 1, (1, 3, 2, 1), "relu"), (40, 2, (3, 11, 9, 4), "gelu"),
                                (64, 3, (6, 30, 20, 8), "gelu"), (64, 6, (2, 30, 20, 8), "relu")]:
        B, X, Y, Z = geometry
        N = B * X * Y * Z
        w = init_encoder_weights(T=11, U=U, L=L, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05,
                                 seed=U + L, spatial_taps=9)
        x = torch.as_tensor(rng.uniform(0.2, 1.0, (B, X, Y, Z, 11)).astype(np.float32), device="cuda")
        g_q = torch.as_tensor(rng.normal(size=(N, 5)).astype(np.float32), device="cuda")
        g_ls = torch.as_tensor(rng.normal(size=(N, 11)).astype(np.float32), device="cuda")
        grads = {}
        for sel in (0, separate):
            ctx = Context(params, True, True)
            ctx.set_kernel_selection(sel)
            ew = EncoderWeights(ctx, 11, U, L, True, -3.0, spatial_taps=9, activation=act).set_from_arrays(w)
            st = TrainState(ctx, ew)
            st.forward_spatial(x)
            grads[sel] = st.backward_spatial(g_q, g_ls, None).clone()
        assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0, (U, L, geometry)
        assert torch.equal(grads[0], grads[separate]), \
            (U, L, geometry, float((grads[0] - grads[separate]).abs().max()))


def test_crop_backward_keeps_its_precision_when_the_deltas_carry_one_over_sum_mask(params):
    """The crop backward's 3 x 3 x 1 backward-data products run on split-f16 operands (conv9h_kernel).  Deltas that carry
    the loss's 1 / sum(mask) sit far under f16's normal range: the kernel lifts them by 2^floor(log2 sum(mask)) on arrival
    and scales its outputs back, both exact.  The gradient with `sums` given must therefore be the gradient without,
    divided by sum(mask), to float32 rounding -- for a sum(mask) of a large crop batch and for one beyond any batch
    (unscaled, deltas of 1e-9 lose their high halves altogether: errors of 1e-2)."""
    from qbold_vi_amd.init import init_encoder_weights
    from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
    rng = np.random.default_rng(17)
    U, L = 60, 2
    B, X, Y, Z = 3, 12, 10, 4
    N = B * X * Y * Z
    w = init_encoder_weights(T=11, U=U, L=L, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=3,
                             spatial_taps=9)
    x = torch.as_tensor(rng.uniform(0.2, 1.0, (B, X, Y, Z, 11)).astype(np.float32), device="cuda")
    g_q = torch.as_tensor(rng.normal(size=(N, 5)).astype(np.float32), device="cuda")
    g_ls = torch.as_tensor(rng.normal(size=(N, 11)).astype(np.float32), device="cuda")
    ctx = Context(params, True, True)
    ew = EncoderWeights(ctx, 11, U, L, True, -3.0, spatial_taps=9).set_from_arrays(w)
    st = TrainState(ctx, ew)
    st.forward_spatial(x)
    plain = st.backward_spatial(g_q, g_ls, None).double().clone()
    for total in (190000.0 * 0.6, 3.0e9):
        sums = torch.tensor([0.0, 0.0, total], dtype=torch.float64, device="cuda")
        st.forward_spatial(x)
        scaled = st.backward_spatial(g_q, g_ls, sums).double().clone() * total
        for name, pieces in ew._slices().items():
            for l, (off, shape) in enumerate(pieces):
                cnt = int(np.prod(shape))
                a, b = plain[off:off + cnt], scaled[off:off + cnt]
                assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-30, \
                    (total, name, l, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize("crops", [False, True])
def test_backward_is_bitwise_reproducible_run_to_run(params, crops):
    """Every reduction of the training step adds in a fixed order (slabs per wave / workgroup summed by slab index, no
    float atomics), and the weight-gradient kernels' running delta scale depends on the data's order only: the same
    inputs give the same gradient bit for bit, launch after launch -- on a voxel batch (one-launch block backward) and on
    crops (matrix-pipe nine-tap kernels, queued slab sums), with deltas that make the scale fall several times."""
    from qbold_vi_amd.init import init_encoder_weights
    from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
    rng = np.random.default_rng(123)
    U, L = 60, 2
    ctx = Context(params, True, True)
    if crops:
        B, X, Y, Z = 4, 14, 10, 8
        N = B * X * Y * Z
        w = init_encoder_weights(T=11, U=U, L=L, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=5,
                                 spatial_taps=9)
        ew = EncoderWeights(ctx, 11, U, L, True, -3.0, spatial_taps=9).set_from_arrays(w)
        x = torch.as_tensor(rng.uniform(0.2, 1.0, (B, X, Y, Z, 11)).astype(np.float32), device="cuda")
    else:
        N = 70001
        w = init_encoder_weights(T=11, U=U, L=L, channelwise_gating=True, resid_init_std=0.3, im_loss_sigma=0.05, seed=5)
        ew = EncoderWeights(ctx, 11, U, L, True, -3.0).set_from_arrays(w)
        x = torch.as_tensor(rng.uniform(0.2, 1.0, (N, 11)).astype(np.float32), device="cuda")
    scale = np.exp(rng.uniform(np.log(1e-8), np.log(1e-3), (N, 1)))
    g_q = torch.as_tensor((rng.normal(size=(N, 5)) * scale).astype(np.float32), device="cuda")
    g_ls = torch.as_tensor((rng.normal(size=(N, 11)) * scale).astype(np.float32), device="cuda")
    st = TrainState(ctx, ew)
    grads = []
    for _ in range(3):
        if crops:
            st.forward_spatial(x)
            grads.append(st.backward_spatial(g_q, g_ls, None).clone())
        else:
            st.forward(x, 2)
            grads.append(st.backward(2, g_q, g_ls).clone())
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])

"""stream 2 with its 3x3x1 'same' convolutions on image crops, the TV smoothness
term, and their gradients -- against the oracle's spatial restatement (model.py:142-174, 726-754)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


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


def make(ctx, U, L, cw, seed=2, gate_offset=-1.0):
    from oracle.oracle import init_weights
    from qbold_vi_amd.ops import EncoderWeights
    w = init_weights(T=11, U=U, L=L, channelwise_gating=cw, seed=seed, taps=9, resid_init_std=0.08)
    rng = np.random.default_rng(seed)
    for k in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
    w["gate_offset"] = gate_offset
    ew = EncoderWeights(ctx, 11, U, L, cw, gate_offset, spatial_taps=9).set_from_arrays(w)
    return w, ew


def crop_batch(oracle32, B, X, Y, Z, seed):
    from oracle.oracle import synth_inputs
    x, _ = synth_inputs(B * X * Y * Z, seed=seed, oracle=oracle32)
    return x.reshape(B, X, Y, Z, 11)


@pytest.mark.parametrize("U,L,cw", [(60, 2, True), (20, 1, False), (33, 1, True)])
def test_spatial_forward_matches_oracle(ctx, oracle32, U, L, cw):
    # (widths that are no multiple of 4 take conv9_kernel's exact-f32 form: conv9h_kernel reads whole float4 quarters)
    from qbold_vi_amd.ops import TrainState
    w, ew = make(ctx, U, L, cw)
    x = crop_batch(oracle32, 3, 7, 6, 2, seed=1)
    st = TrainState(ctx, ew)
    q, ls = st.forward_spatial(dev(x))
    o2, sg = oracle32.encoder_fwd_spatial(w, x)
    assert np.abs(q.cpu().numpy() - o2.reshape(-1, 5)).max() < 3e-5
    assert np.abs(ls.cpu().numpy() - np.log(sg).reshape(-1, 11)).max() < 3e-5
    # the neighbourhood matters: the voxel-wise (centre-tap) result differs ...
    qv, _ = st.forward(dev(x.reshape(-1, 11)), 2)
    assert float((qv - q).abs().max()) > 1e-3
    # ... and equals the fused inference kernels, which take the centre tap of the same weights
    _, q_fused, _ = ctx.encoder_fwd(ew, dev(x.reshape(-1, 11)), want=("out2",))
    assert torch.allclose(qv, q_fused, atol=2e-5)
    # 1x1 crops have no neighbours: spatial == voxel-wise
    q1, _ = st.forward_spatial(dev(x.reshape(-1, 1, 1, 1, 11)))
    assert torch.allclose(q1, qv, atol=1e-6)


def test_smoothness_loss_and_gradient(ctx, oracle64):
    rng = np.random.default_rng(3)
    B, X, Y, Z = 2, 6, 5, 3
    q = rng.normal(size=(B, X, Y, Z, 5)).astype(np.float32)
    mask = (rng.uniform(size=(B, X, Y, Z)) > 0.25).astype(np.float32)
    tv = ctx.smoothness(dev(q), dev(mask))
    want = oracle64.smoothness_loss(q, mask)
    assert abs(float(tv) / mask.sum() - want) < 1e-5 * want
    g = torch.zeros((B * X * Y * Z, 5), device="cuda")
    ctx.smoothness(dev(q), dev(mask), weight=2.5, g_q=g)
    g = g.cpu().numpy().reshape(B, X, Y, Z, 5)
    assert np.all(g[..., [1, 3, 4]] == 0)
    q64 = q.astype(np.float64)
    for ch in (0, 2):
        for idx in [(0, 0, 0, 0), (1, 3, 2, 1), (0, 5, 4, 2), (1, 2, 0, 0)]:
            d = np.zeros_like(q64)
            d[idx + (ch,)] = 1e-5
            fd = (oracle64.smoothness_loss(q64 + d, mask) - oracle64.smoothness_loss(q64 - d, mask)) \
                / 2e-5 * mask.sum() * 2.5
            assert abs(g[idx + (ch,)] - fd) < 1e-3 * (abs(fd) + 1e-2)


@pytest.mark.parametrize("U,L,cw,Z", [(24, 2, True, 2), (20, 1, False, 2), (33, 1, True, 2), (24, 2, True, 4), (60, 2, True, 8)])
def test_spatial_weight_gradient_directional(ctx, oracle32, oracle64, U, L, cw, Z):
    """d/dw [masked-mean NLL + KL + 5 * TV] through the spatial encoder, against central differences
    of the float64 oracle along random weight directions (every tap of the 3x3x1 kernels moves).  Z = 2: the exact
    float32 nine-tap kernels; Z % 4 == 0 with U % 4 == 0: the matrix-pipe kernels the training step runs on --
    conv9h_kernel's backward-data form with its deltas lifted by the sum(mask) the ELBO backward hands over,
    xtd9b_kernel / xtdb_kernel on two f16 halves under the running delta scale, gate_bwd_wg_kernel, the queued slab
    sums -- held to the oracle directly, not to the exact kernels."""
    from oracle.oracle import WEIGHT_NAMES
    from qbold_vi_amd.ops import EncoderWeights, TrainState
    from test_gpu_grad import _perturbed, kl_stopgrad
    w, ew = make(ctx, U, L, cw)
    B, X, Y = 2, 5, 4
    n, S, K, seed, sw = B * X * Y * Z, 2, 4, 9, 5.0
    x = crop_batch(oracle32, B, X, Y, Z, seed=4)
    rng = np.random.default_rng(5)
    mask = (rng.uniform(size=(B, X, Y, Z)) > 0.2).astype(np.float32)
    xf, mf = x.reshape(n, 11), mask.reshape(n)
    prior = oracle32.encoder_fwd(w, xf)[0]
    st = TrainState(ctx, ew)
    q, ls = st.forward_spatial(dev(x))
    sums, gq, gls, _ = ctx.elbo_bwd(dev(xf), dev(mf), q, dev(prior), ls, S, K, seed=seed)
    tv = ctx.smoothness(q.reshape(B, X, Y, Z, 5), dev(mask), weight=sw, g_q=gq)
    grad = st.backward_spatial(gq, gls, sums).cpu().numpy().astype(np.float64)
    zs = oracle32.philox_normals(seed, 0, 0, n, S)
    zk = oracle32.philox_normals(seed, 1, 0, n, K)
    q_fixed = oracle64.encoder_fwd_spatial(w, x)[0].reshape(n, 5)

    def loss(ww):
        o2, sg = oracle64.encoder_fwd_spatial(ww, x)
        qq = o2.reshape(n, 5)
        e = oracle64.elbo(xf, mf, qq, prior, sg.reshape(n, 11), zs, zk)
        kl = kl_stopgrad(oracle64, qq, q_fixed, prior, zk)
        return ((e["nll_v"] * mf).sum() + np.where(mf > 0, kl, 0).sum()) / mf.sum() + \
            sw * oracle64.smoothness_loss(o2, mask)

    for trial in range(4):
        direction = {k: rng.standard_normal(w[k].shape) for k in WEIGHT_NAMES}
        if trial == 1:   # only the spatial kernels
            for k in WEIGHT_NAMES:
                if k not in ("Wr1", "Wr2"):
                    direction[k] *= 0
        if trial == 2:   # only the off-centre taps
            for k in WEIGHT_NAMES:
                direction[k] *= 0 if k not in ("Wr1", "Wr2") else 1
            direction["Wr1"][:, 1, 1] = 0
            direction["Wr2"][:, 1, 1] = 0
        dflat = EncoderWeights(ctx, 11, U, L, cw, -1.0, spatial_taps=9).set_from_arrays(
            {k: direction[k].astype(np.float32) for k in WEIGHT_NAMES}).flat.cpu().numpy().astype(np.float64)
        eps = 3e-5
        fd = (loss(_perturbed(w, direction, eps)) - loss(_perturbed(w, direction, -eps))) / (2 * eps)
        got = float(grad @ dflat)
        assert abs(got - fd) < 1e-2 * (abs(fd) + 0.05), (trial, got, fd)


def test_spatial_gradients_add_over_crops_with_odd_sizes(ctx, oracle32):
    """Crops whose voxel counts are no multiple of 4 or 16 (3 x 5 x 3 x 1 = 45, singles of 15): crops are
    independent, so the gradient of the batch is the sum of the single-crop gradients -- the nine-tap
    weight-gradient and convolution kernels with ragged tails and padded taps on every border."""
    from qbold_vi_amd.ops import TrainState
    w, ew = make(ctx, 60, 2, True)
    B, X, Y, Z = 3, 5, 3, 1
    x = dev(crop_batch(oracle32, B, X, Y, Z, seed=6))
    rng = np.random.default_rng(2)
    gq_all = dev(rng.normal(size=(B, X * Y * Z, 5)).astype(np.float32))
    gls_all = dev(rng.normal(size=(B, X * Y * Z, 11)).astype(np.float32))
    st = TrainState(ctx, ew)

    def grad_of(b0, b1):
        st.forward_spatial(x[b0:b1].contiguous())
        return st.backward_spatial(gq_all[b0:b1].reshape(-1, 5).contiguous(),
                                   gls_all[b0:b1].reshape(-1, 11).contiguous(), None).double().clone()

    full = grad_of(0, B)
    parts = grad_of(0, 1) + grad_of(1, 2) + grad_of(2, 3)
    assert float((full - parts).abs().max()) < 2e-5 * float(full.abs().max())


@pytest.mark.parametrize("B,X,Y,Z,U", [(3, 6, 5, 4, 64), (2, 25, 25, 8, 60), (1, 3, 3, 12, 20), (5, 1, 2, 4, 64)])
def test_nine_tap_weight_gradients_f16_halves_and_bf16_pieces_against_exact_f32(params, oracle32, B, X, Y, Z, U):
    """Layers of U % 4 == 0 units on crops with Z % 4 == 0 take xtd9b_kernel on the matrix pipe: by default every float32
    operand as two f16 halves, the deltas under the wave's running power-of-two scale (all four half products); with
    kernel selection 4194304 (QBOLD_KSEL_DW_BF16_PIECES) as three bfloat16 pieces (six of the nine piece products, no
    scale anywhere); 524288 (QBOLD_KSEL_DW_EXACT_F32) keeps xtd9_kernel's exact float32 products.  Everything else in
    the passes is the same code on the same data, so the gradients may differ by the dropped bits (<= 2^-22 per product)
    and the summation order only.  Head deltas spread over five decades down to 1e-9 and no `sums` is given: the scale
    is found from the data, and has to fall several times on the way."""
    from qbold_vi_amd.ops import Context, TrainState
    N = B * X * Y * Z
    x = dev(crop_batch(oracle32, B, X, Y, Z, seed=9))
    rng = np.random.default_rng(5)
    scale = np.exp(rng.uniform(np.log(1e-9), np.log(1e-4), (N, 1)))
    g_q = dev((rng.normal(size=(N, 5)) * scale).astype(np.float32))
    g_ls = dev((rng.normal(size=(N, 11)) * scale).astype(np.float32))
    grads = {}
    for sel in (0, 4194304, 524288):
        c = Context(params, full_model=True, include_blood=True)
        c.set_grad_node0(False)
        c.set_kernel_selection(sel)
        w, ew = make(c, U, 2, True, seed=4)
        st = TrainState(c, ew)
        st.forward_spatial(x)
        grads[sel] = st.backward_spatial(g_q, g_ls, None).double().clone()
    a = grads[524288]
    for sel in (0, 4194304):
        b = grads[sel]
        assert torch.isfinite(b).all()
        for name, pieces in ew._slices().items():
            for l, (off, shape) in enumerate(pieces):
                cnt = int(np.prod(shape))
                ta, tb = a[off:off + cnt], b[off:off + cnt]
                assert float((ta - tb).abs().max()) <= 2e-6 * float(ta.abs().max()) + 1e-30, \
                    (sel, name, l, float((ta - tb).abs().max()), float(ta.abs().max()))
    b = grads[0]
    # the nine-tap kernels really differ between the selections (the matrix-pipe paths ran)
    assert not torch.equal(a, b) and not torch.equal(grads[4194304], b)

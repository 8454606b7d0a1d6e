"""GPU parity tests: HIP kernels (through the C ABI) against the CPU oracle on identical inputs.

Tolerances follow SURVEY.md 8(d) "Parity gates" and BASELINE.json north_star:
  per-voxel signal <= 1e-5 rel (same F policy), ELBO scalar <= 1e-4 rel,
  posterior means <= 1e-5 abs, variances <= 1e-4 rel (+ tiny abs floor).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(params):
    from qbold_vi_amd.ops import Context
    return Context(params, full_model=True, include_blood=True)


@pytest.fixture(scope="module")
def weights(ctx):
    from oracle.oracle import init_weights
    from qbold_vi_amd.ops import EncoderWeights
    w = init_weights(T=11, U=60, L=2, seed=3)
    # non-trivial biases so that every bias path is exercised
    rng = np.random.default_rng(7)
    for n in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[n] = (rng.standard_normal(w[n].shape) * 0.1).astype(np.float32)
    w["gate_offset"] = -3.0
    ew = EncoderWeights(ctx, 11, 60, 2, True, -3.0).set_from_arrays(w)
    return w, ew


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def rel(a, b, floor=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor)))


def grid_oef_dbv(n=4096, seed=0):
    rng = np.random.default_rng(seed)
    oef = rng.uniform(0.04, 0.84, n)
    dbv = rng.uniform(0.001, 0.201, n)
    y = np.stack([oef, dbv], -1).astype(np.float32)
    y[:4] = [[0.4, 0.12], [0.4, 0.025], [0.04, 0.001], [0.84, 0.201]]
    return y


# ---------------------------------------------------------------------------------------------
def test_signal_fwd_literal_matches_oracle(ctx, oracle32):
    y = grid_oef_dbv(4096)
    ctx.set_tissue_mode("literal")
    try:
        got = ctx.signal_fwd(dev(y)).cpu().numpy()
    finally:
        ctx.set_tissue_mode("table")
    want = oracle32.signal_fwd(y)
    # same algorithm, same float32 semantics: only libm-vs-OCML ulp differences remain
    assert rel(got, want) < 2e-6


def test_signal_fwd_table_matches_oracle(ctx, oracle32, oracle64):
    y = grid_oef_dbv(20000, seed=1)
    got = ctx.signal_fwd(dev(y)).cpu().numpy()
    want = oracle32.signal_fwd(y)
    assert rel(got, want) < 1e-5
    # reported, not gated: distance of both from the float64 evaluation
    truth = oracle64.signal_fwd(y)
    print("table vs f64 truth", rel(got, truth), "oracle32 vs f64 truth", rel(want, truth))


def test_signal_fwd_shapes_and_edges(ctx, oracle32):
    # arbitrary leading dims, V not a multiple of the block, empty input
    y = grid_oef_dbv(3 * 5 * 7 + 1)[:105].reshape(3, 5, 7, 2)
    got = ctx.signal_fwd(dev(y))
    assert got.shape == (3, 5, 7, 11)
    assert rel(got.cpu().numpy(), oracle32.signal_fwd(y)) < 1e-5
    empty = ctx.signal_fwd(torch.empty((0, 2), device="cuda"))
    assert empty.shape == (0, 11)
    one = ctx.signal_fwd(dev(np.array([[0.4, 0.12]], np.float32))).cpu().numpy()
    assert rel(one, oracle32.signal_fwd([[0.4, 0.12]])) < 1e-5
    with pytest.raises(Exception):
        ctx.signal_fwd(torch.zeros((4, 2)))  # CPU tensor: no fallback


def test_signal_loglinear_and_no_blood(params, oracle32):
    from oracle.oracle import Oracle
    from qbold_vi_amd.ops import Context
    y = grid_oef_dbv(2048, seed=2)
    for full, blood in ((False, True), (True, False), (False, False)):
        c = Context(params, full_model=full, include_blood=blood)
        o = Oracle("f32", params, full_model=full, include_blood=blood)
        assert rel(c.signal_fwd(dev(y)).cpu().numpy(), o.signal_fwd(y)) < 1e-5


def test_signal_variable_hct_and_misalignment(params, oracle32):
    """The two options of SignalGenerationLayer that optimal.yaml switches off (signals.py:64-96)."""
    from oracle.oracle import Oracle
    from qbold_vi_amd.ops import Context
    from qbold_vi_amd.signals import SignalGenerationLayer
    rng = np.random.default_rng(12)
    V = 5000
    y = grid_oef_dbv(V, seed=4)
    hct = rng.uniform(0.25, 0.5, V).astype(np.float32)
    alt = np.stack([np.clip(y[:, 0] + 0.15 * rng.standard_normal(V), 0.05, 0.8),
                    np.clip(y[:, 1] + 0.05 * rng.standard_normal(V), 0.002, 0.3)], -1).astype(np.float32)
    idx = np.where(rng.uniform(size=V) < 0.4, rng.integers(4, 10, V), 11).astype(np.int32)
    for full, blood in ((True, True), (False, True), (True, False)):
        c = Context(params, full_model=full, include_blood=blood)
        o = Oracle("f32", params, full_model=full, include_blood=blood)
        for kw in (dict(hct=hct), dict(alt=alt, from_idx=idx), dict(hct=hct, alt=alt, from_idx=idx)):
            want = o.signal_fwd_ex(y, **kw)
            got = c.signal_fwd_ex(dev(y), None if "hct" not in kw else dev(hct),
                                  None if "alt" not in kw else dev(alt),
                                  None if "alt" not in kw else dev(idx)).cpu().numpy()
            assert rel(got, want) < 1e-5, (full, blood, list(kw))
    # the config's haematocrit as a tensor reproduces the plain model to rounding of the constants
    o = oracle32
    same = o.signal_fwd_ex(y, hct=np.full(V, float(params["hct"]), np.float32))
    assert rel(same, o.signal_fwd(y)) < 1e-6
    # misaligned images are exactly the images of the perturbed parameters, the others untouched
    mis = o.signal_fwd_ex(y, alt=alt, from_idx=idx)
    plain, other = o.signal_fwd(y), o.signal_fwd(alt)
    t = np.arange(11)[None, :]
    np.testing.assert_array_equal(mis, np.where(t > idx[:, None], other, plain))
    # the layer draws the augmentation itself: ~prob of the voxels change, only in images 5..10
    layer = SignalGenerationLayer(dict(params, simulate_noise='False'), True, True, misaligned_prob=0.3)
    out = layer(dev(y)).cpu().numpy()
    changed = out != Context(params, True, True).signal_fwd(dev(y)).cpu().numpy()
    assert not changed[:, :5].any()
    assert 0.22 < changed.any(-1).mean() < 0.36
    with pytest.raises(AssertionError):
        SignalGenerationLayer(dict(params, simulate_noise='False'), True, True, variable_hct=True)(dev(y))
    vh = SignalGenerationLayer(dict(params, simulate_noise='False'), True, True, variable_hct=True)
    got = vh(dev(np.concatenate([y, hct[:, None]], -1))).cpu().numpy()
    assert rel(got, o.signal_fwd_ex(y, hct=hct)) < 1e-5


def test_signal_bwd_matches_oracle_jacobian(ctx, oracle32):
    y = grid_oef_dbv(4096, seed=3)
    rng = np.random.default_rng(5)
    g = rng.standard_normal((y.shape[0], 11)).astype(np.float32)
    jac = oracle32.signal_jac(y)  # [V, T, 2], J1-kernel derivative = TF's gradient of bessel_j0
    want = np.einsum("vt,vtk->vk", g.astype(np.float64), jac)
    got = ctx.signal_bwd(dev(y), dev(g)).cpu().numpy()
    scale = np.abs(np.einsum("vt,vtk->vk", np.abs(g).astype(np.float64), np.abs(jac))) + 1e-6
    assert float(np.max(np.abs(got - want) / scale)) < 2e-4
    ctx.set_tissue_mode("literal")
    try:
        got_l = ctx.signal_bwd(dev(y), dev(g)).cpu().numpy()
    finally:
        ctx.set_tissue_mode("table")
    assert float(np.max(np.abs(got_l - want) / scale)) < 2e-4


# ---------------------------------------------------------------------------------------------
def test_encoder_fwd_matches_oracle(ctx, weights, oracle32):
    from oracle.oracle import synth_inputs
    w, ew = weights
    x, _ = synth_inputs(4096 + 17, seed=1, oracle=oracle32)
    o1, o2, sg = ctx.encoder_fwd(ew, dev(x))
    w1, w2, wsg = oracle32.encoder_fwd(w, x)
    assert np.max(np.abs(o1.cpu().numpy() - w1)) < 2e-5
    assert np.max(np.abs(o2.cpu().numpy() - w2)) < 2e-5
    assert rel(sg.cpu().numpy(), wsg) < 2e-5
    # each output on its own (NULL pointers for the others)
    only1, _, _ = ctx.encoder_fwd(ew, dev(x), want=("out1",))
    assert torch.equal(only1, o1)
    _, only2, _ = ctx.encoder_fwd(ew, dev(x), want=("out2",))
    assert torch.equal(only2, o2)


def test_encoder_small_shapes(ctx, oracle32):
    from oracle.oracle import init_weights, synth_inputs
    from qbold_vi_amd.ops import EncoderWeights
    x, _ = synth_inputs(333, seed=4, oracle=oracle32)
    for U, L, cw in ((30, 1, True), (10, 1, False), (64, 2, True), (33, 2, False)):
        w = init_weights(T=11, U=U, L=L, channelwise_gating=cw, seed=U)
        w["gate_offset"] = 0.5
        ew = EncoderWeights(ctx, 11, U, L, cw, 0.5).set_from_arrays(w)
        o1, o2, sg = ctx.encoder_fwd(ew, dev(x))
        w1, w2, wsg = oracle32.encoder_fwd(w, x)
        assert np.max(np.abs(o1.cpu().numpy() - w1)) < 2e-5, (U, L, cw)
        assert np.max(np.abs(o2.cpu().numpy() - w2)) < 2e-5, (U, L, cw)
        assert rel(sg.cpu().numpy(), wsg) < 2e-5, (U, L, cw)


# ---------------------------------------------------------------------------------------------
def make_q(n, seed=0):
    rng = np.random.default_rng(seed)
    q = np.empty((n, 5), np.float32)
    q[:, 0] = rng.normal(-0.2, 0.8, n)
    q[:, 1] = rng.normal(0.0, 0.7, n)
    q[:, 2] = rng.normal(-1.5, 0.8, n)
    q[:, 3] = rng.normal(0.0, 0.7, n)
    q[:, 4] = rng.normal(0.0, 1.0, n)
    return q


def test_reparam_and_nlogp(ctx, oracle32):
    n = 10000
    q, z = make_q(n, 1), np.random.default_rng(2).standard_normal((n, 2)).astype(np.float32)
    z[:8] *= 6.0  # drive some samples into the clip of model.py:395
    y = ctx.reparam(dev(q), dev(z)).cpu().numpy()
    want = oracle32.reparam(q, z)
    assert np.max(np.abs(y - want)) < 2e-6
    p = make_q(n, 3)
    got = ctx.logit_mvn_nlogp(dev(want), dev(p)).cpu().numpy()
    ref = oracle32.logit_mvn_nlogp(want, p)
    # terms of order 1e2..1e3 cancel inside nlogp: float32 leaves ~1e-7 of the largest term
    assert np.max(np.abs(got - ref) / (np.abs(ref) + 1.0)) < 1e-4


def test_posterior_moments(ctx, oracle32):
    n, ns = 5000, 200
    q = make_q(n, 4)
    z = np.random.default_rng(5).standard_normal((n, ns, 2)).astype(np.float32)
    m, v = ctx.posterior_moments(dev(q), ns, z=dev(z))
    wm, wv = oracle32.moments(q, z)
    assert np.max(np.abs(m.cpu().numpy()[:, :2] - wm[:, :2])) < 1e-5
    assert rel(m.cpu().numpy()[:, 2], wm[:, 2], 1e-3) < 1e-4
    assert rel(v.cpu().numpy(), wv, 1e-12) < 1e-4      # SURVEY 8(d) gate: variances 1e-4 rel (measured 1e-5)
    # Philox stream: integer part bit-identical to the oracle's, normals to a few ulp
    m2, v2 = ctx.posterior_moments(dev(q), 20, seed=11, voxel0=123)
    zp = oracle32.philox_normals(11, 2, 123, n, 20)
    wm2, wv2 = oracle32.moments(q, zp)
    assert np.max(np.abs(m2.cpu().numpy()[:, :2] - wm2[:, :2])) < 1e-5
    assert rel(v2.cpu().numpy(), wv2, 1e-12) < 1e-3    # 20 draws, normals equal to a few ulp only


# ---------------------------------------------------------------------------------------------
def elbo_inputs(oracle32, w, n, seed):
    from oracle.oracle import synth_inputs
    x, _ = synth_inputs(n, seed=seed, oracle=oracle32)
    prior, q, sigma = oracle32.encoder_fwd(w, x)
    rng = np.random.default_rng(seed + 100)
    mask = (rng.uniform(size=n) > 0.2).astype(np.float32)
    return x, mask, q, prior, sigma


@pytest.mark.parametrize("S,K", [(1, 70), (32, 70), (3, 5)])
def test_elbo_explicit_eps(ctx, weights, oracle32, S, K):
    w, _ = weights
    n = 2048 + 5
    x, mask, q, prior, sigma = elbo_inputs(oracle32, w, n, 1)
    rng = np.random.default_rng(9)
    zs = rng.standard_normal((n, S, 2)).astype(np.float32)
    zk = rng.standard_normal((n, K, 2)).astype(np.float32)
    want = oracle32.elbo(x, mask, q, prior, sigma, zs, zk)
    sums, nk = ctx.elbo_fwd(dev(x), dev(mask), dev(q), dev(prior), dev(sigma), S, K, dev(zs), dev(zk))
    sums = sums.cpu().numpy()
    nk = nk.cpu().numpy()
    # Per-voxel NLL: at one draw per voxel the float32 oracle itself is ~1.2e-4 away from exact
    # arithmetic (rounding noise of its 129-term float32 Simpson sum, amplified by 1/sigma = 20),
    # so the bound against it is 2e-4 there; against the float64 evaluation of the same float32
    # semantics (node 0 removed) the kernel is held to 1e-4 at every S (measured 3e-5 / 1e-5).
    assert rel(nk[:, 0], want["nll_v"], 1.0) < (2e-4 if S == 1 else 1e-4)
    from oracle.oracle import Oracle
    o64 = Oracle("f64", oracle32.params, node0_zero=True)
    try:
        truth = o64.elbo(x, mask, q, prior, sigma, zs, zk)
    finally:
        o64.lib.qbo_set_node0_zero(0)
    assert rel(nk[:, 0], truth["nll_v"], 1.0) < 1e-4
    assert np.max(np.abs(nk[:, 1] - want["kl_v"]) / (np.abs(want["kl_v"]) + 1.0)) < 1e-4
    assert sums[2] == want["sums"][2]
    elbo = sums[0] / sums[2] + sums[1] / sums[2]
    assert abs(elbo - want["elbo"]) / abs(want["elbo"]) < 1e-4
    assert abs(elbo - truth["elbo"]) / abs(truth["elbo"]) < 1e-5


def test_elbo_philox_matches_oracle_stream(ctx, weights, oracle32):
    w, _ = weights
    n, S, K, seed, v0 = 1500, 32, 70, 1234, 1000003
    x, mask, q, prior, sigma = elbo_inputs(oracle32, w, n, 2)
    zs = oracle32.philox_normals(seed, 0, v0, n, S)
    zk = oracle32.philox_normals(seed, 1, v0, n, K)
    want = oracle32.elbo(x, mask, q, prior, sigma, zs, zk)
    sums, nk = ctx.elbo_fwd(dev(x), dev(mask), dev(q), dev(prior), dev(sigma), S, K, seed=seed,
                            voxel0=v0)
    sums = sums.cpu().numpy()
    elbo = (sums[0] + sums[1]) / sums[2]
    assert abs(elbo - want["elbo"]) / abs(want["elbo"]) < 1e-4
    assert rel(nk.cpu().numpy()[:, 0], want["nll_v"], 1.0) < 2e-4


def test_elbo_literal_mode(ctx, weights, oracle32):
    w, _ = weights
    n, S, K = 512, 2, 4
    x, mask, q, prior, sigma = elbo_inputs(oracle32, w, n, 3)
    rng = np.random.default_rng(1)
    zs = rng.standard_normal((n, S, 2)).astype(np.float32)
    zk = rng.standard_normal((n, K, 2)).astype(np.float32)
    want = oracle32.elbo(x, mask, q, prior, sigma, zs, zk)
    ctx.set_tissue_mode("literal")
    try:
        sums, nk = ctx.elbo_fwd(dev(x), dev(mask), dev(q), dev(prior), dev(sigma), S, K, dev(zs), dev(zk))
    finally:
        ctx.set_tissue_mode("table")
    assert rel(nk.cpu().numpy()[:, 0], want["nll_v"], 1.0) < 5e-5


def test_vi_fwd_fused_matches_oracle(ctx, weights, oracle32):
    """BASELINE config 1: optimal.yaml, 4k synthetic voxels x 11 tau (S=1 and S=32, K=70)."""
    from oracle.oracle import synth_inputs
    w, ew = weights
    n, K, seed = 4096, 70, 1
    x, _ = synth_inputs(n, seed=1, oracle=oracle32)
    prior, q_want, sigma = oracle32.encoder_fwd(w, x)
    mask = np.ones(n, np.float32)
    for S in (1, 32):
        zs = oracle32.philox_normals(seed, 0, 0, n, S)
        zk = oracle32.philox_normals(seed, 1, 0, n, K)
        want = oracle32.elbo(x, mask, q_want, prior, sigma, zs, zk)
        sums, q, nk = ctx.vi_fwd(ew, dev(x), dev(mask), dev(prior), S, K, seed=seed)
        sums = sums.cpu().numpy()
        assert np.max(np.abs(q.cpu().numpy() - q_want)) < 2e-5
        elbo = (sums[0] + sums[1]) / sums[2]
        print("S", S, "elbo gpu", elbo, "oracle", want["elbo"])
        assert abs(elbo - want["elbo"]) / abs(want["elbo"]) < 1e-4
        assert rel(nk.cpu().numpy()[:, 0], want["nll_v"], 1.0) < 5e-4
        # fused == unfused on the same encoder outputs and the same Philox stream
        s2, nk2 = ctx.elbo_fwd(dev(x), dev(mask), q, dev(prior),
                               ctx.encoder_fwd(ew, dev(x), want=("sigma",))[2], S, K, seed=seed)
        assert torch.allclose(nk, nk2, rtol=1e-5, atol=1e-5)


def test_vi_fwd_sharding_invariance(ctx, weights, oracle32):
    """Philox counters are keyed by the global voxel index: two half-shards == one full batch."""
    from oracle.oracle import synth_inputs
    w, ew = weights
    n = 3000
    x, _ = synth_inputs(n, seed=5, oracle=oracle32)
    prior = dev(oracle32.encoder_fwd(w, x)[0])
    xd = dev(x)
    full, qf, nkf = ctx.vi_fwd(ew, xd, None, prior, 4, 10, seed=3, voxel0=0)
    h = 1472
    a, qa, nka = ctx.vi_fwd(ew, xd[:h], None, prior[:h], 4, 10, seed=3, voxel0=0)
    b, qb, nkb = ctx.vi_fwd(ew, xd[h:], None, prior[h:], 4, 10, seed=3, voxel0=h)
    assert torch.equal(torch.cat([nka, nkb]), nkf)
    assert torch.equal(torch.cat([qa, qb]), qf)
    assert torch.allclose(a + b, full, rtol=1e-12)


def test_vi_fwd_ragged_and_empty_batches(ctx, weights, oracle32):
    """Batch sizes around the 16-voxel wave tile and the 256-voxel block pass, a single voxel, and
    the empty batch: each voxel's outputs do not depend on what else is in the batch."""
    from oracle.oracle import synth_inputs
    w, ew = weights
    x, _ = synth_inputs(300, seed=8, oracle=oracle32)
    prior = dev(oracle32.encoder_fwd(w, x)[0])
    xd = dev(x)
    mask = dev((np.random.default_rng(3).uniform(size=300) > 0.3).astype(np.float32))
    full, qf, nkf = ctx.vi_fwd(ew, xd, mask, prior, 3, 5, seed=9)
    for n in (1, 2, 15, 16, 17, 255, 257):
        s, q, nk = ctx.vi_fwd(ew, xd[:n], mask[:n], prior[:n], 3, 5, seed=9)
        assert torch.equal(q, qf[:n]) and torch.equal(nk, nkf[:n]), n
        want = torch.stack([(nkf[:n, 0].double() * mask[:n].double()).sum(),
                            nkf[:n, 1].double()[mask[:n] > 0].sum(), mask[:n].double().sum()])
        assert torch.allclose(s, want, rtol=1e-6, atol=1e-9), n
    s0, q0, nk0 = ctx.vi_fwd(ew, xd[:0], mask[:0], prior[:0], 3, 5, seed=9)
    assert s0.tolist() == [0.0, 0.0, 0.0] and q0.shape == (0, 5) and nk0.shape == (0, 2)
    e0, _ = ctx.elbo_fwd(xd[:0], mask[:0], qf[:0], prior[:0], torch.empty((0, 11), device="cuda"), 3, 5, seed=9)
    assert e0.tolist() == [0.0, 0.0, 0.0]
    # odd S / K (half-used Philox calls) agree with the oracle too
    for S, K in ((1, 1), (5, 3)):
        sums, q, nk = ctx.vi_fwd(ew, xd[:64], None, prior[:64], S, K, seed=2)
        qn, sg = q.cpu().numpy(), oracle32.encoder_fwd(w, x[:64])[2]
        want = oracle32.elbo(x[:64], np.ones(64, np.float32), qn, prior[:64].cpu().numpy(), sg,
                             oracle32.philox_normals(2, 0, 0, 64, S), oracle32.philox_normals(2, 1, 0, 64, K))
        assert abs(float((sums[0] + sums[1]) / sums[2]) - want["elbo"]) < 1e-4 * abs(want["elbo"]), (S, K)


def test_vi_fwd_random_configurations(params, oracle32):
    """A sweep over seeded random cases of the fused kernel: encoder width / depth / gating, draw
    counts, masks, voxel offsets, gate offsets and weight scales -- ELBO within the north-star
    tolerance of the oracle on each."""
    from oracle.oracle import init_weights, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights
    ctx = Context(params, True, True)
    rng = np.random.default_rng(2024)
    for case in range(12):
        U = int(rng.choice([8, 17, 32, 47, 60, 64]))
        L = int(rng.choice([1, 2]))
        cw = bool(rng.integers(0, 2))
        S, K = int(rng.integers(1, 40)), int(rng.integers(0, 80))
        n = int(rng.integers(1, 900))
        v0 = int(rng.integers(0, 2 ** 40))
        seed = int(rng.integers(0, 2 ** 62))
        goff = float(rng.uniform(-3, 1))
        w = init_weights(T=11, U=U, L=L, channelwise_gating=cw, seed=case, resid_init_std=float(rng.uniform(0.02, 0.2)))
        for nm in ("b0", "bc", "br1", "br2", "bg", "bf"):
            w[nm] = (rng.standard_normal(w[nm].shape) * 0.1).astype(np.float32)
        w["gate_offset"] = goff
        ew = EncoderWeights(ctx, 11, U, L, cw, goff).set_from_arrays(w)
        x, _ = synth_inputs(n, seed=case, oracle=oracle32)
        prior, q_want, sigma = oracle32.encoder_fwd(w, x)
        mask = (rng.uniform(size=n) > rng.uniform(0, 0.6)).astype(np.float32)
        if mask.sum() == 0:
            mask[0] = 1.0
        sums, q, nk = ctx.vi_fwd(ew, dev(x), dev(mask), dev(prior), S, K, seed=seed, voxel0=v0)
        assert np.abs(q.cpu().numpy() - q_want).max() < 3e-5, (case, U, L, cw)
        want = oracle32.elbo(x, mask, q_want, prior, sigma, oracle32.philox_normals(seed, 0, v0, n, S),
                             oracle32.philox_normals(seed, 1, v0, n, max(K, 1)) if K else np.zeros((n, 1, 2), np.float32))
        sums = sums.cpu().numpy()
        got = (sums[0] + sums[1]) / sums[2]
        ref = want["elbo"] if K else want["sums"][0] / want["sums"][2]
        assert abs(got - ref) < 1e-4 * abs(ref), (case, U, L, cw, S, K, n, got, ref)


def test_vi_fwd_random_configurations_24_tau(params):
    """The same sweep on the reference's 24-tau protocol (spin-echo index 7: seven mirrored tau pairs in the
    likelihood loop), plus per-voxel agreement of the fused kernel with the unfused ELBO kernel and -- in
    literal mode -- with the oracle's literal Simpson sums (exactly mirrored pairs share a tissue factor)."""
    from oracle.oracle import Oracle, init_weights, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights
    p = dict(params, tau_start="-0.028", tau_end="0.065", tau_step="0.004")
    orc = Oracle("f32", p)
    ctx = Context(p, True, True)
    rng = np.random.default_rng(77)
    for case in range(8):
        U, L, cw = int(rng.choice([17, 32, 60, 64])), int(rng.choice([1, 2])), bool(rng.integers(0, 2))
        S, K = int(rng.integers(1, 12)), int(rng.integers(0, 30))
        n = int(rng.integers(1, 500))
        v0, seed = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 62))
        w = init_weights(T=24, U=U, L=L, channelwise_gating=cw, seed=case)
        w["gate_offset"] = -2.0
        ew = EncoderWeights(ctx, 24, U, L, cw, -2.0).set_from_arrays(w)
        x, _ = synth_inputs(n, p, seed=case, oracle=orc)
        prior, q_want, sigma = orc.encoder_fwd(w, x)
        mask = (rng.uniform(size=n) > 0.3).astype(np.float32)
        mask[0] = 1.0
        zs = orc.philox_normals(seed, 0, v0, n, S)
        zk = orc.philox_normals(seed, 1, v0, n, max(K, 1)) if K else np.zeros((n, 1, 2), np.float32)
        for mode in ("table", "literal") if case < 3 else ("table",):
            ctx.set_tissue_mode(mode)
            sums, q, nk = ctx.vi_fwd(ew, dev(x), dev(mask), dev(prior), S, K, seed=seed, voxel0=v0)
            want = orc.elbo(x, mask, q_want, prior, sigma, zs, zk)
            sums = sums.cpu().numpy()
            got = (sums[0] + sums[1]) / sums[2]
            ref = want["elbo"] if K else want["sums"][0] / want["sums"][2]
            assert abs(got - ref) < 1e-4 * abs(ref), (case, mode, U, L, cw, S, K, n, got, ref)
            s2, nk2 = ctx.elbo_fwd(dev(x), dev(mask), q, dev(prior), dev(sigma), S, K, seed=seed, voxel0=v0)
            assert torch.allclose(nk, nk2, rtol=2e-4, atol=2e-4), (case, mode)
        ctx.set_tissue_mode("table")


def test_protocol_whose_spin_echo_image_is_not_at_tau_zero(params):
    """tau_start = -0.017 s with 8 ms steps: model.py:95 still picks image 2 as the normaliser, but its tau is
    -1 ms, so the grid does not mirror about it -- the compile-time spin-echo kernels must take their general
    path (no table-free spin-echo signal, no shared pairs).  Fused, unfused and literal against the oracle."""
    from oracle.oracle import Oracle, init_weights, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights
    p = dict(params, tau_start="-0.017", tau_end="0.071", tau_step="0.008")
    orc = Oracle("f32", p)
    ctx = Context(p, True, True)
    assert ctx.T == orc.T == 11 and ctx.se_idx == 2
    w = init_weights(T=11, U=60, L=2, seed=9)
    w["gate_offset"] = -3.0
    ew = EncoderWeights(ctx, 11, 60, 2, True, -3.0).set_from_arrays(w)
    n, S, K, seed = 700, 6, 11, 5
    x, _ = synth_inputs(n, p, seed=3, oracle=orc)
    prior, q_want, sigma = orc.encoder_fwd(w, x)
    mask = np.ones(n, np.float32)
    want = orc.elbo(x, mask, q_want, prior, sigma, orc.philox_normals(seed, 0, 0, n, S),
                    orc.philox_normals(seed, 1, 0, n, K))
    for mode in ("table", "literal"):
        ctx.set_tissue_mode(mode)
        sums, q, nk = ctx.vi_fwd(ew, dev(x), dev(mask), dev(prior), S, K, seed=seed)
        sums = sums.cpu().numpy()
        assert abs((sums[0] + sums[1]) / sums[2] - want["elbo"]) < 1e-4 * abs(want["elbo"]), mode
        s2, nk2 = ctx.elbo_fwd(dev(x), dev(mask), q, dev(prior), dev(sigma), S, K, seed=seed)
        assert torch.allclose(nk, nk2, rtol=2e-4, atol=2e-4), mode
        assert np.max(np.abs(nk.cpu().numpy()[:, 0] - want["nll_v"]) / (np.abs(want["nll_v"]) + 1.0)) < 2e-4, mode


def test_full_size_properties_one_million_voxels(ctx, weights, oracle32, params):
    """BASELINE config 2 at its full size (1,048,576 voxels x 11 tau, S=32, K=70), through the
    properties that do not need the oracle on every voxel: determinism, shard additivity, masking,
    the sums as a checksum of the per-voxel outputs -- and the oracle on a window in the middle."""
    from qbold_vi_amd.signals import SignalGenerationLayer
    w, ew = weights
    N, S, K, seed = 1 << 20, 32, 70, 11
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    y = torch.stack([torch.rand(N, generator=g, device="cuda") * 0.7 + 0.08,
                     torch.rand(N, generator=g, device="cuda") * 0.1 + 0.005], -1)
    x = SignalGenerationLayer(dict(params, simulate_noise='True'), True, True)(y)
    prior = ctx.encoder_fwd(ew, x, want=("out1",))[0]
    mask = (torch.rand(N, generator=g, device="cuda") > 0.25).float()
    sums, q, nk = ctx.vi_fwd(ew, x, mask, prior, S, K, seed=seed)
    sums2, q2, nk2 = ctx.vi_fwd(ew, x, mask, prior, S, K, seed=seed)
    assert torch.equal(sums, sums2) and torch.equal(nk, nk2) and torch.equal(q, q2)   # bitwise repeatable
    assert bool(torch.isfinite(nk).all()) and bool(torch.isfinite(sums).all())
    # sums are the masked checksum of the per-voxel outputs
    nkd, md = nk.double(), mask.double()
    assert abs(float((nkd[:, 0] * md).sum()) / float(sums[0]) - 1) < 1e-8    # float32 lane partials (16 tiles per lane)
    assert abs(float(nkd[:, 1][mask > 0].sum()) / float(sums[1]) - 1) < 1e-8    # float32 lane partials of a cancelling sum
    assert float(sums[2]) == float(md.sum())
    # four shards keyed by their first global voxel == the whole batch
    parts, acc = [], torch.zeros_like(sums)
    for r in range(4):
        sl = slice(r * (N // 4), (r + 1) * (N // 4))
        s_r, _, nk_r = ctx.vi_fwd(ew, x[sl], mask[sl], prior[sl], S, K, seed=seed, voxel0=sl.start)
        parts.append(nk_r)
        acc += s_r
    assert torch.equal(torch.cat(parts), nk)
    assert torch.allclose(acc, sums, rtol=1e-7, atol=0)   # reduction order differs across shards (float32 lane partials)
    # an all-zero mask contributes nothing; no mask == mask of ones
    z, _, _ = ctx.vi_fwd(ew, x[:4096], torch.zeros(4096, device="cuda"), prior[:4096], S, K, seed=seed)
    assert z.tolist() == [0.0, 0.0, 0.0]
    a, _, _ = ctx.vi_fwd(ew, x[:4096], None, prior[:4096], S, K, seed=seed)
    b, _, _ = ctx.vi_fwd(ew, x[:4096], torch.ones(4096, device="cuda"), prior[:4096], S, K, seed=seed)
    assert torch.equal(a, b)
    # the oracle on a 2048-voxel window keyed by its global position
    v0, n = 777_000, 2048
    xs, ms, ps = (t[v0:v0 + n].cpu().numpy() for t in (x, mask, prior))
    _, q_want, sigma = oracle32.encoder_fwd(w, xs)
    want = oracle32.elbo(xs, ms, q_want, ps, sigma, oracle32.philox_normals(seed, 0, v0, n, S),
                         oracle32.philox_normals(seed, 1, v0, n, K))
    got = nk[v0:v0 + n].cpu().numpy()
    assert rel(got[:, 0], want["nll_v"], 1.0) < 5e-4
    assert np.max(np.abs(got[:, 1] - want["kl_v"]) / (np.abs(want["kl_v"]) + 1.0)) < 5e-4
    elbo = (float((got[:, 0].astype(np.float64) * ms).sum()) + float(got[:, 1].astype(np.float64)[ms > 0].sum())) / ms.sum()
    assert abs(elbo - want["elbo"]) / abs(want["elbo"]) < 1e-4


def test_bf16_encoder_mode(ctx, weights, oracle32):
    """BASELINE config 5 ("bf16 forward / fp32 ELBO accum, tolerance re-stated"): the encoder's
    matrix products on bfloat16-rounded operands with float32 accumulation; sampling, forward
    signal model and ELBO sums stay float32.
    Tolerances (re-stated for this mode):
      * vs the oracle with the same bf16 operand rounding: |dq| < 2e-3 (an activation within an
        accumulation-order ulp of a bf16 rounding boundary may round the other way), ELBO 1e-4 rel
        given identical q / sigma;
      * vs the float32 path: |dq| < 0.03 in logit units, sigma within 1e-2 rel (bf16 carries 8
        mantissa bits through five 64-wide layers; measured 5e-3 / 2e-3), ELBO within 1e-3 rel
        (measured 7e-5)."""
    from oracle.oracle import synth_inputs
    from qbold_vi_amd.ops import EncoderWeights
    w, ew32 = weights
    ew = EncoderWeights(ctx, 11, 60, 2, True, w["gate_offset"], precision="bf16").set_from_arrays(w)
    n, S, K, seed = 4096, 32, 70, 3
    x, _ = synth_inputs(n, seed=9, oracle=oracle32)
    xd = dev(x)
    o1, o2, sg = (t.cpu().numpy() for t in ctx.encoder_fwd(ew, xd))
    oracle32.set_encoder_bf16(True)
    try:
        w1, w2, wsg = oracle32.encoder_fwd(w, x)
    finally:
        oracle32.set_encoder_bf16(False)
    f1, f2, fsg = oracle32.encoder_fwd(w, x)
    print("bf16 kernel vs bf16 oracle:", np.abs(o2 - w2).max(), " bf16 vs f32:", np.abs(o2 - f2).max(),
          np.abs(sg / fsg - 1).max())
    assert np.abs(o1 - w1).max() < 2e-3 and np.abs(o2 - w2).max() < 2e-3
    assert np.abs(sg / wsg - 1).max() < 2e-3
    assert 1e-5 < np.abs(o2 - f2).max() < 0.03         # it really is a different arithmetic
    assert np.abs(sg / fsg - 1).max() < 1e-2
    # fused ELBO: float32 sampling on the bf16 encoder's outputs
    prior = dev(f1)
    sums, q, nk = ctx.vi_fwd(ew, xd, None, prior, S, K, seed=seed)
    assert np.abs(q.cpu().numpy() - o2).max() < 1e-6    # same encoder arithmetic as encoder_fwd
    want = oracle32.elbo(x, np.ones(n, np.float32), q.cpu().numpy(), f1, sg,
                         oracle32.philox_normals(seed, 0, 0, n, S), oracle32.philox_normals(seed, 1, 0, n, K))
    elbo = float((sums[0] + sums[1]) / sums[2])
    assert abs(elbo - want["elbo"]) / abs(want["elbo"]) < 1e-4
    s32, _, _ = ctx.vi_fwd(ew32, xd, None, prior, S, K, seed=seed)
    elbo32 = float((s32[0] + s32[1]) / s32[2])
    print("ELBO bf16", elbo, "f32", elbo32, "rel", abs(elbo / elbo32 - 1))
    assert abs(elbo / elbo32 - 1) < 1e-3
    # the literal / generic paths reproduce float32 semantics only
    ctx.set_tissue_mode("literal")
    try:
        with pytest.raises(RuntimeError, match="QBOLD_ENC_BF16"):
            ctx.vi_fwd(ew, xd[:64], None, prior[:64], 1, 1, seed=seed)
    finally:
        ctx.set_tissue_mode("table")


def test_24_tau_protocol(params):
    """The reference's second acquisition protocol (24 taus from -0.028 s in 4 ms steps,
    signals.py:120-121): forward model, encoder, fused ELBO and head gradients."""
    from oracle.oracle import Oracle, init_weights, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights
    p = dict(params, tau_start="-0.028", tau_end="0.065", tau_step="0.004")
    orc = Oracle("f32", p)
    ctx = Context(p, full_model=True, include_blood=True)
    assert ctx.T == orc.T == 24 and ctx.se_idx == 7
    y = grid_oef_dbv(3000, seed=8)
    assert rel(ctx.signal_fwd(dev(y)).cpu().numpy(), orc.signal_fwd(y)) < 1e-5
    w = init_weights(T=24, U=60, L=2, seed=4)
    w["gate_offset"] = -3.0
    ew = EncoderWeights(ctx, 24, 60, 2, True, -3.0).set_from_arrays(w)
    n, S, K, seed = 1200, 5, 9, 3
    x, _ = synth_inputs(n, p, seed=8, noise=False, oracle=orc)
    x = (x * (1 + 0.01 * np.random.default_rng(0).standard_normal(x.shape))).astype(np.float32)
    prior, q_want, sigma = orc.encoder_fwd(w, x)
    o1, o2, sg = ctx.encoder_fwd(ew, dev(x))
    assert np.max(np.abs(o2.cpu().numpy() - q_want)) < 2e-5 and rel(sg.cpu().numpy(), sigma) < 2e-5
    mask = np.ones(n, np.float32)
    want = orc.elbo(x, mask, q_want, prior, sigma, orc.philox_normals(seed, 0, 0, n, S),
                    orc.philox_normals(seed, 1, 0, n, K))
    sums, q, nk = ctx.vi_fwd(ew, dev(x), dev(mask), dev(prior), S, K, seed=seed)
    sums = sums.cpu().numpy()
    assert abs((sums[0] + sums[1]) / sums[2] - want["elbo"]) < 1e-4 * abs(want["elbo"])
    s2, nk2 = ctx.elbo_fwd(dev(x), dev(mask), q, dev(prior), sg, S, K, seed=seed)
    assert torch.allclose(nk, nk2, rtol=1e-4, atol=1e-4)
    s3, gq, gls, nk3 = ctx.elbo_bwd(dev(x), dev(mask), q, dev(prior), torch.log(sg), S, K, seed=seed)
    assert torch.allclose(nk3, nk2, rtol=1e-4, atol=1e-4) and bool(torch.isfinite(gq).all())
    # the backward's one-lane-per-voxel mapping (S <= 2, the training defaults) against the forward kernel's values
    s4, nk4 = ctx.elbo_fwd(dev(x), dev(mask), q, dev(prior), sg, 1, 70, seed=seed)
    s5, gq1, gls1, nk5 = ctx.elbo_bwd(dev(x), dev(mask), q, dev(prior), torch.log(sg), 1, 70, seed=seed)
    assert torch.allclose(nk5, nk4, rtol=1e-4, atol=1e-4) and bool(torch.isfinite(gq1).all() and torch.isfinite(gls1).all())


def test_wide_encoder_shapes_and_edges(params, oracle32):
    """The weight-streaming encoder on the 11-tau grid (one padded K chunk in the first layer), with
    voxel counts around its 128- / 256-voxel block passes, and its argument checks."""
    from oracle.oracle import init_weights, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights
    ctx = Context(params, True, True)
    for U, L, ns in ((256, 1, (1, 127, 129, 1000)), (128, 2, (31, 256, 257))):
        w = init_weights(T=11, U=U, L=L, seed=U + 1)
        rb = np.random.default_rng(U)
        for nm in ("b0", "bc", "br1", "br2", "bg", "bf"):
            w[nm] = (rb.standard_normal(w[nm].shape) * 0.1).astype(np.float32)
        w["gate_offset"] = 0.5
        ew = EncoderWeights(ctx, 11, U, L, True, 0.5).set_from_arrays(w)
        assert ew.wide
        for n in ns:
            x, _ = synth_inputs(n, seed=n, oracle=oracle32)
            want = oracle32.encoder_fwd(w, x)
            got = ctx.encoder_fwd(ew, dev(x))
            for a, b, tol in zip(got, want, (5e-5, 5e-5, None)):
                if tol is None:
                    assert rel(a.cpu().numpy(), b) < 5e-5
                else:
                    assert np.max(np.abs(a.cpu().numpy() - b)) < tol
    # shared (non channel-wise) gating is not built for the streaming path: falls back to layer-wise
    ew1 = EncoderWeights(ctx, 11, 128, 1, False, 0.0)
    assert not ew1.wide


def test_config3_wide_encoder_64_taus(params):
    """BASELINE config 3 shapes (T = 64 taus, encoder width 256, SURVEY H6 tau grid) through the
    weight-streaming MFMA encoder (U = 256 / 128; other widths take the layer-wise f32 GEMMs) and
    the any-T ELBO kernel, against the oracle at small N."""
    from oracle.oracle import Oracle, init_weights, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights
    p = dict(params, tau_start="-0.015", tau_end="0.065", tau_step="0.00125")
    orc = Oracle("f32", p)
    ctx = Context(p, full_model=True, include_blood=True)
    assert ctx.T == orc.T == 64 and ctx.se_idx == 12
    n, S, K, seed = 700, 4, 6, 5
    x, _ = synth_inputs(n, p, seed=3, noise=False, oracle=orc)
    x = (x * (1 + 0.01 * np.random.default_rng(1).standard_normal(x.shape))).astype(np.float32)
    assert rel(ctx.signal_fwd(dev(synth_inputs(50, p, seed=1, noise=False, oracle=orc)[1])).cpu().numpy(),
               orc.signal_fwd(synth_inputs(50, p, seed=1, noise=False, oracle=orc)[1])) < 1e-5
    for U, L in ((256, 2), (128, 3), (100, 3)):
        w = init_weights(T=64, U=U, L=L, seed=U)
        rb = np.random.default_rng(U)
        for nm in ("b0", "bc", "br1", "br2", "bg", "bf"):
            w[nm] = (rb.standard_normal(w[nm].shape) * 0.1).astype(np.float32)
        w["gate_offset"] = -3.0
        ew = EncoderWeights(ctx, 64, U, L, True, -3.0).set_from_arrays(w)
        assert ew.wide == (U in (128, 256))
        prior, q_want, sigma = orc.encoder_fwd(w, x)
        o1, o2, sg = ctx.encoder_fwd(ew, dev(x))
        assert np.max(np.abs(o1.cpu().numpy() - prior)) < 5e-5
        assert np.max(np.abs(o2.cpu().numpy() - q_want)) < 5e-5
        assert rel(sg.cpu().numpy(), sigma) < 5e-5
        mask = (np.random.default_rng(2).uniform(size=n) > 0.1).astype(np.float32)
        want = orc.elbo(x, mask, q_want, prior, sigma, orc.philox_normals(seed, 0, 0, n, S),
                        orc.philox_normals(seed, 1, 0, n, K))
        sums, q, nk = ctx.vi_fwd(ew, dev(x), dev(mask), dev(prior), S, K, seed=seed)
        sums = sums.cpu().numpy()
        assert abs((sums[0] + sums[1]) / sums[2] - want["elbo"]) < 1e-4 * abs(want["elbo"])
        assert rel(nk.cpu().numpy()[:, 0], want["nll_v"], 1.0) < 5e-4
        # explicit normals through the any-T kernel as well
        rng = np.random.default_rng(4)
        zs = rng.standard_normal((n, S, 2)).astype(np.float32)
        zk = rng.standard_normal((n, K, 2)).astype(np.float32)
        want2 = orc.elbo(x, mask, q_want, prior, sigma, zs, zk)
        s2, nk2 = ctx.elbo_fwd(dev(x), dev(mask), dev(q_want), dev(prior), dev(sigma), S, K, dev(zs), dev(zk))
        # 64 residuals at sigma ~ 0.05 with an untrained posterior: |r| ~ 4, so the table's ~2e-6
        # signal error is amplified to a few 1e-4 of a per-voxel NLL of order 1e3
        assert rel(nk2.cpu().numpy()[:, 0], want2["nll_v"], 1.0) < 5e-4
        # per-voxel KL = mean of log q - log p, two O(10) terms that cancel to O(0.1): float32 noise
        assert np.max(np.abs(nk2.cpu().numpy()[:, 1] - want2["kl_v"]) / (np.abs(want2["kl_v"]) + 1.0)) < 5e-4


def _config3_params(params):
    return dict(params, tau_start="-0.015", tau_end="0.065", tau_step="0.00125")


def test_one_launch_wide_encoder_matches_oracle_and_layerwise(params):
    """wide_fused_kernel (the whole stream-2 encoder of U = 256 in one launch, activations in registers) against
    the oracle and against the layer-wise weight-streaming kernels, on both instantiated tau counts (one and four
    first-layer k-steps), one and two blocks, voxel counts around its 128-voxel passes and a multi-pass batch;
    non-zero biases everywhere, a weight scale spread over four decades (the per-op power-of-two scaling)."""
    from oracle.oracle import Oracle, init_weights
    from qbold_vi_amd.ops import Context, EncoderWeights
    for T, p in ((11, params), (64, _config3_params(params))):
        orc = Oracle("f32", p)
        ctx = Context(p, True, True)
        assert ctx.T == T
        for L in (1, 2):
            w = init_weights(T=T, U=256, L=L, seed=10 * T + L)
            rb = np.random.default_rng(T + L)
            for nm in ("b0", "bc", "br1", "br2", "bg", "bf", "bs"):
                w[nm] = (w[nm] + rb.standard_normal(w[nm].shape) * 0.1).astype(np.float32)
            w["Wr1"] = (w["Wr1"] * 10.0 ** rb.uniform(-3, 0.3, size=(L, 1, 256))).astype(np.float32)
            w["gate_offset"] = -1.0
            ew = EncoderWeights(ctx, T, 256, L, True, -1.0).set_from_arrays(w)
            assert ew.fused_wide
            for n in (1, 127, 129, 1000, 70001):
                g = torch.Generator(device="cuda")
                g.manual_seed(n)
                x = torch.rand((n, T), generator=g, device="cuda") * 0.6 + 0.15
                ctx.force_layerwise_wide = True
                _, q0, s0 = ctx.encoder_fwd(ew, x, want=("out2", "sigma"))
                ctx.force_layerwise_wide = False
                _, q1, s1 = ctx.encoder_fwd(ew, x, want=("out2", "sigma"))
                assert float((q0 - q1).abs().max()) < 5e-6 and float(((s0 - s1).abs() / s0).max()) < 5e-6, (T, L, n)
                if n <= 1000:
                    _, q_want, s_want = orc.encoder_fwd(w, x.cpu().numpy())
                    assert np.max(np.abs(q1.cpu().numpy() - q_want)) < 2e-5, (T, L, n)
                    assert rel(s1.cpu().numpy(), s_want) < 2e-5, (T, L, n)
            # re-packing after an update is picked up (per-op scales are recomputed from the new weights)
            w2 = dict(w, W0=(w["W0"] * 3.0).astype(np.float32))
            ew.set_from_arrays(w2)
            x = torch.rand((300, T), device="cuda") * 0.6 + 0.15
            _, q_want, _ = orc.encoder_fwd(w2, x.cpu().numpy())
            assert np.max(np.abs(ctx.encoder_fwd(ew, x, want=("out2",))[1].cpu().numpy() - q_want)) < 2e-5


def test_config3_full_size_properties(params):
    """BASELINE config 3 at its full size (1,048,576 voxels x 64 tau, encoder width 256, S = 32, K = 70) through
    qbold_vi_fwd's two-launch wide path: bitwise determinism, the sums as the checksum of the per-voxel outputs,
    shard additivity with global-voxel Philox keys, the oracle on a window -- and a 9 M-voxel batch whose signal
    and log-sigma tensors pass 2^31 bytes (64-bit offsets), checked by the oracle on a window at its tail."""
    from oracle.oracle import Oracle, init_weights
    from qbold_vi_amd.ops import Context, EncoderWeights
    from qbold_vi_amd.signals import SignalGenerationLayer
    p = _config3_params(params)
    orc = Oracle("f32", p)
    ctx = Context(p, True, True)
    w = init_weights(T=64, U=256, L=2, seed=3)
    rb = np.random.default_rng(7)
    for nm in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[nm] = (rb.standard_normal(w[nm].shape) * 0.1).astype(np.float32)
    w["gate_offset"] = -3.0
    ew = EncoderWeights(ctx, 64, 256, 2, True, -3.0).set_from_arrays(w)
    S, K, seed = 32, 70, 11

    def batch(N, gseed):
        g = torch.Generator(device="cuda")
        g.manual_seed(gseed)
        y = torch.stack([torch.rand(N, generator=g, device="cuda") * 0.7 + 0.08,
                         torch.rand(N, generator=g, device="cuda") * 0.1 + 0.005], -1)
        x = SignalGenerationLayer(dict(p, simulate_noise='False'), True, True)(y)   # (noise model: 11 / 24 taus only)
        x = x * (1 + 0.01 * torch.randn(x.shape, generator=g, device="cuda"))
        return x.contiguous(), (torch.rand(N, generator=g, device="cuda") > 0.25).float()

    def window(x, mask, prior, nk, q, v0, n):
        xs, ms, ps = (t[v0:v0 + n].cpu().numpy() for t in (x, mask, prior))
        _, q_want, sigma = orc.encoder_fwd(w, xs)
        assert np.max(np.abs(q[v0:v0 + n].cpu().numpy() - q_want)) < 2e-5
        want = orc.elbo(xs, ms, q_want, ps, sigma, orc.philox_normals(seed, 0, v0, n, S),
                        orc.philox_normals(seed, 1, v0, n, K))
        got = nk[v0:v0 + n].cpu().numpy()
        assert rel(got[:, 0], want["nll_v"], 1.0) < 5e-4
        assert np.max(np.abs(got[:, 1] - want["kl_v"]) / (np.abs(want["kl_v"]) + 1.0)) < 5e-4

    N = 1 << 20
    x, mask = batch(N, 5)
    prior = ctx.encoder_fwd(ew, x, want=("out1",))[0]
    sums, q, nk = ctx.vi_fwd(ew, x, mask, prior, S, K, seed=seed)
    sums2, q2, nk2 = ctx.vi_fwd(ew, x, mask, prior, S, K, seed=seed)
    assert torch.equal(sums, sums2) and torch.equal(nk, nk2) and torch.equal(q, q2)   # bitwise repeatable
    assert bool(torch.isfinite(nk).all()) and bool(torch.isfinite(sums).all())
    nkd, md = nk.double(), mask.double()
    assert abs(float((nkd[:, 0] * md).sum()) / float(sums[0]) - 1) < 1e-8    # float32 lane partials (16 tiles per lane)
    assert abs(float(nkd[:, 1][mask > 0].sum()) / float(sums[1]) - 1) < 1e-8    # float32 lane partials of a cancelling sum
    assert float(sums[2]) == float(md.sum())
    parts, acc = [], torch.zeros_like(sums)
    for a, b in ((0, 300_001), (300_001, 700_000), (700_000, N)):   # ragged shards: partial passes in the middle
        s_r, _, nk_r = ctx.vi_fwd(ew, x[a:b].contiguous(), mask[a:b].contiguous(), prior[a:b].contiguous(), S, K,
                                  seed=seed, voxel0=a)
        parts.append(nk_r)
        acc += s_r
    assert torch.equal(torch.cat(parts), nk)           # per voxel: bit for bit, whatever the sharding
    # the three sums: a lane adds its voxels' terms in float32 before the float64 block / grid reduction, and
    # ragged shards regroup the voxels over lanes -- float32 rounding of sixteen-term partial sums
    assert torch.allclose(acc, sums, rtol=1e-7, atol=0)
    window(x, mask, prior, nk, q, 777_000, 256)
    del x, mask, prior, q, nk, q2, nk2, parts
    # 9 M voxels: x and log sigma are 2.3 GB each
    N = 9_000_000
    x, mask = batch(N, 6)
    assert x.numel() * 4 > 2 ** 31
    prior = ctx.encoder_fwd(ew, x[N - 4096:].contiguous(), want=("out1",))[0]
    prior = torch.cat([torch.zeros((N - 4096, 5), device="cuda"), prior])
    prior[:, 1] = torch.where(prior[:, 1] == 0, torch.full_like(prior[:, 1], -0.3), prior[:, 1])
    sums, q, nk = ctx.vi_fwd(ew, x, mask, prior, S, K, seed=seed)
    assert bool(torch.isfinite(sums).all()) and float(sums[2]) == float(mask.double().sum())
    window(x, mask, prior, nk, q, N - 300, 300)      # the last, partial pass; byte offsets beyond 2^31
    window(x, mask, prior, nk, q, 8_500_000, 128)


def test_split_operand_range(ctx, weights, oracle32):
    """The f16 operand split of the MFMA encoders against the float32 oracle at the ends of its range
    (include/qbold_hip.h, QBOLD_ENC_F32): (1) hidden activations of several 1e5 -- beyond f16's 65504 -- poison the
    voxel's outputs: NaN heads from qbold_encoder_fwd, non-finite sums from qbold_vi_fwd (the status channel; left
    alone, inf - inf = NaN in the accumulators and relu's v_max_f32 would turn it into a finite, wrong number), and
    vi_fwd(range_check=True) recomputes on the exact-float32 layer-wise path, which matches the oracle; the same
    for a WEIGHT beyond 65504; (2) activations just inside the range keep float32-grade parity; (3) hidden
    activations of 1e-5 (f16 subnormals in both halves) amplified back to O(1) by the heads keep an absolute error
    of the 2.9e-11 resolution times the amplification."""
    from oracle.oracle import synth_inputs
    from qbold_vi_amd.ops import EncoderWeights
    w, _ = weights
    n, S, K, seed = 512, 4, 8, 3
    x, _ = synth_inputs(n, seed=12, oracle=oracle32)
    mask = np.ones(n, np.float32)

    def scaled(sc_first, sc_block, sc_out):
        w2 = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in w.items()}
        w2["W0"], w2["b0"] = w["W0"] * sc_first, w["b0"] * sc_first   # first-layer activations scale with sc_first
        w2["Wc"], w2["bc"] = w["Wc"].copy(), w["bc"].copy()
        w2["Wc"][0] *= sc_block                                        # ... block 0's skip path with sc_block on top
        w2["bc"][0] *= sc_block
        w2["Wf"], w2["Ws"] = w["Wf"] * sc_out, w["Ws"] * sc_out       # heads bring the scale back
        return w2

    def run(w2, range_check):
        ew = EncoderWeights(ctx, 11, 60, 2, True, -3.0).set_from_arrays(w2)
        prior, q_want, sigma = oracle32.encoder_fwd(w2, x)
        want = oracle32.elbo(x, mask, q_want, prior, sigma, oracle32.philox_normals(seed, 0, 0, n, S),
                             oracle32.philox_normals(seed, 1, 0, n, K))
        sums, q, nk = ctx.vi_fwd(ew, dev(x), dev(mask), dev(prior), S, K, seed=seed, range_check=range_check)
        return ew, want, q_want, sums.cpu().numpy(), q.cpu().numpy()

    # (1) overflow by ACTIVATIONS: weights of a few hundred, block-0 activations of several 1e5
    big = scaled(1e3, 1e3, 1e-6)
    assert max(np.abs(big[k]).max() for k in ("W0", "Wc")) < 6e4
    assert np.abs(oracle32.encoder_fwd(big, x)[1]).max() < 50        # the float32 reference is perfectly finite
    ew, want, q_want, sums, q = run(big, range_check=False)
    assert not np.isfinite(sums).any() or not np.isfinite(sums[:2]).all()   # status: non-finite sums
    o2 = ctx.encoder_fwd(ew, dev(x), want=("out2",))[1]
    assert int(torch.isnan(o2).any(1).sum()) > n // 2                # poisoned heads, not clamped numbers
    before = getattr(ctx, "range_fallbacks", 0)
    ew, want, q_want, sums, q = run(big, range_check=True)
    assert ctx.range_fallbacks == before + 1 and np.isfinite(sums).all()
    assert np.max(np.abs(q - q_want)) < 1e-3 * max(1.0, np.abs(q_want).max())   # float32 GEMMs of 1e5-sized terms
    assert abs((sums[0] + sums[1]) / sums[2] - want["elbo"]) < 1e-3 * abs(want["elbo"])
    # ... and by a WEIGHT the pack kernel cannot split
    hot = scaled(3e5, 1.0, 1.0 / 3e5)
    assert np.abs(hot["W0"]).max() > 65504
    ew, want, q_want, sums, q = run(hot, range_check=False)
    assert not np.isfinite(sums[:2]).all()
    # (2) just inside: activations of a few 1e4
    mid = scaled(20.0, 1e3, 5e-5)
    ew, want, q_want, sums, q = run(mid, range_check=False)
    assert np.isfinite(sums).all() and np.max(np.abs(q - q_want)) < 5e-5 * max(1.0, np.abs(q_want).max())
    assert abs((sums[0] + sums[1]) / sums[2] - want["elbo"]) < 1e-4 * abs(want["elbo"])
    # (3) tiny activations (1e-5 and less: below f16's smallest normal 6.1e-5), amplified by 1e5 in the heads
    tiny = scaled(1e-5, 1.0, 1e5)      # (head weights of ~2e4: an amplification of 1e6 would leave the WEIGHT range)
    for k in ("bc", "br1", "br2"):     # every hidden activation scales with 1e-5 (the layers are positively homogeneous)
        tiny[k] = w[k] * 1e-5
    assert np.abs(oracle32.encoder_fwd(tiny, x)[1]).max() < 50
    ew, want, q_want, sums, q = run(tiny, range_check=False)
    assert np.isfinite(sums).all()
    # absolute resolution 2^-35 per operand, 60 terms, x 2e4 head weights: a few 1e-5 on an O(1) head output
    assert np.max(np.abs(q - q_want)) < 2e-4


@pytest.mark.parametrize("T", [49, 50, 52, 60, 63, 12, 16])
def test_one_launch_wide_encoder_on_partly_filled_tau_tiles(params, T):
    """The one-launch wide encoder on tau counts that fill their tiles only partly: 49..63 taus (four first-layer
    k-steps whose last is partly padding, two log-sigma tiles whose second holds T - 32 live rows) and 12 / 16 (one
    k-step).  T % 4 != 0 takes the dword head stores and scalar signal loads; T % 4 == 0 the 16-byte forms with
    their per-handshake vmcnt allowances, and with T % 8 == 4 the signal window that is fetched from T - 8 and
    picked apart in convert_x (a round-2 bug: the row's last four taus used to be dropped).  Heads against the
    oracle, ragged voxel counts around the 128-voxel pass."""
    from oracle.oracle import Oracle, init_weights
    from qbold_vi_amd.ops import Context, EncoderWeights
    p = dict(params, tau_start="-0.010", tau_end=str(-0.010 + 0.001 * T - 0.0005), tau_step="0.001")
    orc = Oracle("f32", p)
    ctx = Context(p, True, True)
    assert ctx.T == orc.T == T and ctx.se_idx == 10
    w = init_weights(T=T, U=256, L=2, seed=T)
    rng = np.random.default_rng(T)
    for nm in ("b0", "bc", "br1", "br2", "bg", "bf", "bs"):
        w[nm] = (w[nm] + 0.1 * rng.standard_normal(w[nm].shape)).astype(np.float32)
    w["gate_offset"] = -3.0
    ew = EncoderWeights(ctx, T, 256, 2, True, -3.0).set_from_arrays(w)
    assert ew.fused_wide
    for n in (1, 127, 130, 517):
        x = rng.uniform(0.2, 1.0, (n, T)).astype(np.float32)
        _, q_want, sg_want = orc.encoder_fwd(w, x)
        _, q, sg = ctx.encoder_fwd(ew, torch.as_tensor(x, device="cuda"), want=("out2", "sigma"))
        assert np.abs(q.cpu().numpy() - q_want).max() < 5e-5, (T, n)
        assert (np.abs(sg.cpu().numpy() - sg_want) / sg_want).max() < 5e-5, (T, n)


def test_per_tau_table_against_the_x_indexed_table(ctx, weights, oracle32, params):
    """The sampling fast path reads the tissue integral from a per-tau OEF-indexed table (one segment index per DRAW,
    MEASUREMENTS.md 4.3); QBOLD_KSEL_X_TABLE selects the round-1/2 form with one x-indexed lookup per (draw, tau).  Both against
    the oracle on the same Philox stream, fused and unfused, 11 and 24 taus."""
    from oracle.oracle import Oracle, init_weights, synth_inputs
    from qbold_vi_amd.ops import Context, EncoderWeights
    w, ew = weights
    n, S, K, seed = 4096, 32, 70, 13
    x, _ = synth_inputs(n, seed=21, oracle=oracle32)
    prior, q_want, sigma = oracle32.encoder_fwd(w, x)
    mask = (np.random.default_rng(2).uniform(size=n) > 0.2).astype(np.float32)
    want = oracle32.elbo(x, mask, q_want, prior, sigma, oracle32.philox_normals(seed, 0, 0, n, S),
                         oracle32.philox_normals(seed, 1, 0, n, K))
    got = {}
    try:
        for sel in (0, 8):
            ctx.set_kernel_selection(sel)
            _, _, nk_f = ctx.vi_fwd(ew, dev(x), dev(mask), dev(prior), S, K, seed=seed)
            _, nk_u = ctx.elbo_fwd(dev(x), dev(mask), dev(q_want), dev(prior), dev(sigma), S, K, seed=seed)
            got[sel] = (nk_f.cpu().numpy(), nk_u.cpu().numpy())
    finally:
        ctx.set_kernel_selection(0)
    for sel, (nk_f, nk_u) in got.items():
        for nk in (nk_f, nk_u):
            assert rel(nk[:, 0], want["nll_v"], 1.0) < 2e-4, sel
            assert rel(nk[:, 1], want["kl_v"], 1.0) < 2e-4, sel
    assert not np.array_equal(got[0][0], got[8][0])                    # two different kernels did run
    assert rel(got[0][0][:, 0], got[8][0][:, 0], 1.0) < 5e-5           # ... and agree far inside the oracle gate
    assert rel(got[0][1][:, 0], got[8][1][:, 0], 1.0) < 5e-5
    # 24 taus: the x-indexed table under either selection (the per-tau table is built for the 11-tau protocol)
    p24 = dict(params, tau_start="-0.028", tau_end="0.065", tau_step="0.004")
    o24 = Oracle("f32", p24)
    c24 = Context(p24, True, True)
    w24 = init_weights(T=24, U=60, L=2, seed=5)
    w24["gate_offset"] = -3.0
    e24 = EncoderWeights(c24, 24, 60, 2, True, -3.0).set_from_arrays(w24)
    x24 = o24.signal_fwd(np.stack([np.random.default_rng(3).uniform(0.1, 0.7, 1024),
                                   np.random.default_rng(4).uniform(0.005, 0.1, 1024)], -1).astype(np.float32))
    x24 = (x24 * (1 + 0.01 * np.random.default_rng(5).standard_normal(x24.shape))).astype(np.float32)
    p1, q24, s24 = o24.encoder_fwd(w24, x24)
    want24 = o24.elbo(x24, np.ones(1024, np.float32), q24, p1, s24, o24.philox_normals(seed, 0, 0, 1024, 8),
                      o24.philox_normals(seed, 1, 0, 1024, 20))
    for sel in (0, 8):
        c24.set_kernel_selection(sel)
        _, _, nk = c24.vi_fwd(e24, dev(x24), None, dev(p1), 8, 20, seed=seed)
        assert rel(nk.cpu().numpy()[:, 0], want24["nll_v"], 1.0) < 2e-4, sel


def test_whitened_kl_draws_against_the_general_form(ctx, oracle32):
    """The KL draws on in-kernel normals run in whitened form (|d + M z|^2 - |z|^2, elbo_core.h) while the clip of the
    logits cannot bind; explicit normals and posteriors wide enough to reach the clip take the general loop.  Same
    stream, both forms, against each other and the oracle -- including posteriors whose draws ARE clipped."""
    n, K, seed = 2048, 70, 31
    rng = np.random.default_rng(17)
    q = make_q(n, seed=5)
    prior = make_q(n, seed=6)
    x = np.abs(rng.normal(0.4, 0.05, (n, 11))).astype(np.float32) + 0.1
    sigma = np.full((n, 11), 0.05, np.float32)
    z = oracle32.philox_normals(seed, 1, 0, n, K)
    assert np.abs(z).max() <= 4.8549     # QB_Z_MAX: sqrt(-2 ln 2^-17), the stream's sixteen-bit radius
    want = oracle32.kl_samples(q, prior, z)
    white = ctx.kl_fwd(dev(q), dev(prior), K=K, seed=seed).cpu().numpy()                 # stand-alone KL kernel
    _, nk_w = ctx.elbo_fwd(dev(x), None, dev(q), dev(prior), dev(sigma), 2, K, seed=seed)            # whitened
    _, nk_g = ctx.elbo_fwd(dev(x), None, dev(q), dev(prior), dev(sigma), 2, K, seed=seed, zk=dev(z),
                           zs=dev(oracle32.philox_normals(seed, 0, 0, n, 2)))                          # general
    # q and prior far apart (KL up to ~2e3): the float32 general form -- the reference's own arithmetic, oracle and
    # kernels alike -- carries ~5e-4 of cancellation error there; the whitened form never forms the large logits and
    # sits at float32 rounding of the result.  Measured against the float64 evaluation: general 7.7e-4, whitened 2e-6.
    from oracle.oracle import Oracle
    want64 = Oracle("f64").kl_samples(q, prior, z)
    assert rel(white, want, 1.0) < 1e-3 and rel(nk_g[:, 1].cpu().numpy(), want, 1.0) < 1e-3     # general form
    assert rel(nk_w[:, 1].cpu().numpy(), want64, 1.0) < 2e-5                                     # whitened form
    assert rel(nk_w[:, 1].cpu().numpy(), nk_g[:, 1].cpu().numpy(), 1.0) < 2e-3
    assert torch.allclose(nk_w[:, 0], nk_g[:, 0], rtol=2e-4, atol=2e-4)                  # same likelihood draws
    # wide posteriors far from the centre: draws beyond the clip at +-13.8155 (model.py:393-396)
    qw = q.copy()
    qw[:, 0] = rng.choice([-11.0, 11.0], n)
    qw[:, 1] = 3.0            # s_o -> 2: e^s = 7.4
    qw[::2, 2] = -12.0
    a = qw[:, 0:1] + z[:, :, 0] * np.exp(3 * np.tanh(qw[:, 1:2]) - 1)
    assert (np.abs(a) > 13.8155).mean() > 0.2                                            # the clip does bind
    want_w = oracle32.kl_samples(qw, prior, z)
    _, nk_c = ctx.elbo_fwd(dev(x), None, dev(qw), dev(prior), dev(sigma), 2, K, seed=seed)
    # general form on both sides, KL up to ~1.6e4: float32 cancellation of the large logits on either side (the
    # whitened formula without the clip would be off by O(1) relative here)
    assert rel(nk_c[:, 1].cpu().numpy(), want_w, 1.0) < 5e-3
    # draw counts that are no multiple of four: the untaken draws of the last Philox call (1, 2 and 3 of them), and
    # fewer calls than lanes per voxel
    for k_odd in (7, 6, 5, 70, 3, 1):
        want_o = Oracle("f64").kl_samples(q, prior, oracle32.philox_normals(seed, 1, 0, n, k_odd))
        _, nk_o = ctx.elbo_fwd(dev(x), None, dev(q), dev(prior), dev(sigma), 2, k_odd, seed=seed)
        assert rel(nk_o[:, 1].cpu().numpy(), want_o, 1.0) < 2e-5, k_odd

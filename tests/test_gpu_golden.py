"""The HIP path against the COMMITTED fixtures (tests/golden/restatement_goldens.npz, written by
tests/golden/make_restatement_goldens.py) -- no live oracle involved, so these numbers are the same
on every box.  The fixtures are restatement goldens (parity unpinned, see the generator's header)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "restatement_goldens.npz")
NAMES = ("W0", "b0", "Wc", "bc", "Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Wf", "bf", "Ws", "bs")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def ctx(params):
    from qbold_vi_amd.ops import Context
    return Context(params, True, True)


def test_forward_model_fixture(ctx, g):
    np.testing.assert_array_equal(ctx.taus, g["taus"])
    got = ctx.signal_fwd(dev(g["sig_pts"])).cpu().numpy()
    np.testing.assert_allclose(got, g["sig_f32"], rtol=1e-5)
    F, _ = ctx.table_eval(g["F_x"].astype(np.float32))
    # the float32 129-term sum carries ~4e-6 of absolute rounding noise that the smooth table does not
    # reproduce; it enters the signal as exp(-dbv F) with dbv <= 0.2 (hence the 1e-5 on signals above)
    np.testing.assert_allclose(F, g["F_f32"], rtol=2e-6, atol=6e-6)
    ex = ctx.signal_fwd_ex(dev(g["ex_y"]), dev(g["ex_hct"]), dev(g["ex_alt"]), dev(g["ex_idx"])).cpu().numpy()
    np.testing.assert_allclose(ex, g["ex_sig"], rtol=1e-5)


def test_encoder_and_elbo_fixture(ctx, g):
    from qbold_vi_amd.ops import EncoderWeights
    ew = EncoderWeights(ctx, 11, 60, 2, True, -3.0).set_from_arrays({k: g["w_" + k] for k in NAMES})
    o1, o2, sg = ctx.encoder_fwd(ew, dev(g["x"]))
    assert np.abs(o1.cpu().numpy() - g["prior"]).max() < 2e-5
    assert np.abs(o2.cpu().numpy() - g["q"]).max() < 2e-5
    np.testing.assert_allclose(sg.cpu().numpy(), g["sigma"], rtol=2e-5)
    S, K = g["zs"].shape[1], g["zk"].shape[1]
    sums, nk = ctx.elbo_fwd(dev(g["x"]), dev(g["mask"]), dev(g["q"]), dev(g["prior"]), dev(g["sigma"]), S, K,
                            dev(g["zs"]), dev(g["zk"]))
    sums = sums.cpu().numpy()
    np.testing.assert_allclose(sums[2], g["sums"][2])
    elbo = (sums[0] + sums[1]) / sums[2]
    assert abs(elbo - float(g["elbo"])) < 1e-4 * abs(float(g["elbo"]))   # the north-star tolerance
    nk = nk.cpu().numpy()
    assert np.max(np.abs(nk[:, 0] - g["nll_v"]) / (np.abs(g["nll_v"]) + 1.0)) < 1e-4
    assert np.max(np.abs(nk[:, 1] - g["kl_v"]) / (np.abs(g["kl_v"]) + 1.0)) < 1e-4
    m, v = ctx.posterior_moments(dev(g["q"]), g["zm"].shape[1], z=dev(g["zm"]))
    assert np.abs(m.cpu().numpy()[:, :2] - g["means"][:, :2]).max() < 1e-5
    np.testing.assert_allclose(v.cpu().numpy(), g["vars"], rtol=1e-4, atol=1e-12)
    z = ctx.normals(8, 6, stream_id=0, seed=1, voxel0=5).cpu().numpy()
    # integer Philox stream bit-identical; v_sin / v_cos are absolute-error instructions (~2e-5 near a zero)
    np.testing.assert_allclose(z, g["philox_z"], rtol=1e-5, atol=4e-5)


def test_crop_and_wls_fixture(ctx, g):
    from qbold_vi_amd.ops import EncoderWeights, TrainState
    ew = EncoderWeights(ctx, 11, 12, 2, True, -1.0, spatial_taps=9).set_from_arrays({k: g["w9_" + k] for k in NAMES})
    st = TrainState(ctx, ew, optimiser_state=False)
    q, ls = st.forward_spatial(dev(g["crop"]))
    assert np.abs(q.cpu().numpy() - g["sp_q"].reshape(-1, 5)).max() < 2e-5
    np.testing.assert_allclose(np.exp(ls.cpu().numpy()), g["sp_sigma"].reshape(-1, 11), rtol=2e-5)
    tv = ctx.smoothness(dev(g["sp_q"]), dev(g["crop_mask"]))
    assert abs(float(tv[0]) / g["crop_mask"].sum() - float(g["tv"])) < 1e-5 * float(g["tv"])
    out = ctx.wls_fit(dev((g["x"][:32].astype(np.float64) * 200.0).astype(np.float32))).cpu().numpy()
    np.testing.assert_allclose(out[:, 2:3], g["wls_r2p"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out[:, 1:2], g["wls_dbv"], rtol=1e-4, atol=2e-6)

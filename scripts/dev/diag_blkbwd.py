"""Dev: where do non-finite values appear in the fused block backward?"""
import configparser, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qbold_vi_amd.init import init_encoder_weights
from qbold_vi_amd.ops import Context, EncoderWeights, TrainState
cfg = configparser.ConfigParser(); cfg.read(os.path.join(ROOT, "config")); p = dict(cfg["DEFAULT"])
T, L, U, N = 11, 2, 60, 16 * 37 + 5
w = init_encoder_weights(T=T, U=U, L=L, channelwise_gating=True, resid_init_std=0.3, im_loss_sigma=0.05, seed=4)
rng = np.random.default_rng(11)
x = torch.as_tensor(rng.uniform(0.2, 1.0, (N, T)).astype(np.float32), device="cuda")
for fill, small in ((0.0, False), (float("nan"), False), (0.0, True), (float("nan"), True)):
    scale = np.exp(rng.uniform(np.log(1e-9), np.log(1e-4), (N, 1))) if small else 1.0
    g_q = torch.as_tensor((rng.normal(size=(N, 5)) * scale).astype(np.float32), device="cuda")
    g_ls = torch.as_tensor((rng.normal(size=(N, T)) * scale).astype(np.float32), device="cuda")
    ctx = Context(p, True, True)
    ew = EncoderWeights(ctx, T, U, L, True, -3.0).set_from_arrays(w)
    st = TrainState(ctx, ew)
    st.workspace(N).fill_(fill)
    q, ls = st.forward(x, 2)
    grad = st.backward(2, g_q, g_ls).clone()
    ws = st.workspace(N)
    base = 2 + 5 * L
    print(f"fill={fill} small={small}: grad finite {bool(torch.isfinite(grad).all())}")
    for name, pieces in ew._slices().items():
        for l, (off, shape) in enumerate(pieces):
            g = grad[off:off + int(np.prod(shape))]
            if not torch.isfinite(g).all():
                print("   non-finite:", name, l, int((~torch.isfinite(g)).sum()), "of", g.numel())
    for k, nm in enumerate(("dA(dT)", "dB", "dC", "dD", "dE")):
        s = ws[(base + k) * N * 64:(base + k + 1) * N * 64].reshape(N, 64)
        bad = ~torch.isfinite(s)
        if bad.any():
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print("   ", nm, "non-finite rows", rows[:8].tolist(), "n", len(rows), "cols", cols[:8].tolist(), len(cols))
    import ctypes as C
    lib = ctx.lib
    etot = int(lib.qbold_encoder_packed_floats(C.byref(ew.shape)))
    off = (base + 5) * N * 64 + 8 * 512 * 4160 + 4096
    imf = ws[off:off + etot]; imb = ws[off + etot:off + 2 * etot]; wt = ws[off + 2 * etot: off + 2 * etot + ew.num_params]
    ew.packed_ptr()
    print("   img_f finite", bool(torch.isfinite(imf).all()), "== packed", bool((imf == ew.packed).all()),
          "img_b finite", bool(torch.isfinite(imb).all()), "wt finite", bool(torch.isfinite(wt).all()),
          "nonfinite idx img_b", (~torch.isfinite(imb)).nonzero().flatten()[:6].tolist(), int((~torch.isfinite(imb)).sum()))

"""Dev check: one-launch wide encoder vs the layer-wise wide encoder on the GPU (no oracle)."""
import configparser, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qbold_vi_amd.ops import Context, EncoderWeights
from qbold_vi_amd.init import init_encoder_weights
cfg = configparser.ConfigParser(); cfg.read(os.path.join(ROOT, "config")); params = dict(cfg["DEFAULT"])
for T in (11, 64):
    p = dict(params)
    if T == 64:
        p.update(tau_start="-0.015", tau_end="0.065", tau_step="0.00125")
    ctx = Context(p, True, True)
    for L in (1, 2):
        w = init_encoder_weights(T=T, U=256, L=L, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=3)
        rb = np.random.default_rng(5)
        for nm in ("b0", "bc", "br1", "br2", "bg", "bf"):
            w[nm] = (rb.standard_normal(np.shape(w[nm])) * 0.1).astype(np.float32)
        ew = EncoderWeights(ctx, T, 256, L, True, -1.0).set_from_arrays(w)
        assert ew.fused_wide
        for n in (1, 127, 129, 1000, 70000):
            g = torch.Generator(device="cuda"); g.manual_seed(n)
            x = torch.rand((n, T), generator=g, device="cuda") * 0.5 + 0.2
            ctx.force_layerwise_wide = True
            _, q0, s0 = ctx.encoder_fwd(ew, x, want=("out2", "sigma"))
            ctx.force_layerwise_wide = False
            _, q1, s1 = ctx.encoder_fwd(ew, x, want=("out2", "sigma"))
            torch.cuda.synchronize()
            dq = (q0 - q1).abs().max().item(); ds = ((s0 - s1).abs() / s0.abs()).max().item()
            bad = ((q0 - q1).abs().amax(1) > 1e-4).nonzero().flatten()[:8].tolist()
            print(f"T={T} L={L} n={n}: max|dq|={dq:.3e} max rel dsigma={ds:.3e} bad voxels {bad}", flush=True)

"""Dev: the one-launch wide encoder against the layer-wise wide path and the oracle for several tau counts."""
import configparser, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, init_weights
from qbold_vi_amd.ops import Context, EncoderWeights
cfg = configparser.ConfigParser(); cfg.read(os.path.join(ROOT, "config")); params = dict(cfg["DEFAULT"])
for T in (64, 56, 60, 52, 50, 49, 63, 16, 12, 11):
    p = dict(params, tau_start="-0.010", tau_end=str(-0.010 + 0.001 * T - 0.0005), tau_step="0.001")
    orc = Oracle("f32", p); ctx = Context(p, True, True)
    w = init_weights(T=T, U=256, L=2, seed=T); w['gate_offset'] = -3.0
    ew = EncoderWeights(ctx, T, 256, 2, True, -3.0).set_from_arrays(w)
    rng = np.random.default_rng(T)
    n = 130
    x = rng.uniform(0.2, 1.0, (n, T)).astype(np.float32)
    _, q_want, sg_want = orc.encoder_fwd(w, x)
    xd = torch.as_tensor(x, device="cuda")
    _, q, sg = ctx.encoder_fwd(ew, xd, want=("out2", "sigma"))
    ctx.force_layerwise_wide = True
    _, q2, sg2 = ctx.encoder_fwd(ew, xd, want=("out2", "sigma"))
    ctx.force_layerwise_wide = False
    print(f"T={T:2d} se={ctx.se_idx} fused_wide={ew.fused_wide}: |q-oracle| {np.abs(q.cpu().numpy()-q_want).max():.2e}  |q_layerwise-oracle| {np.abs(q2.cpu().numpy()-q_want).max():.2e}"
          f"  sigma rel {np.abs(sg.cpu().numpy()/sg_want-1).max():.2e} / {np.abs(sg2.cpu().numpy()/sg_want-1).max():.2e}")

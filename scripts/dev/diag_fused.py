"""Dev: head / bias mapping of wide_fused_kernel with zero weights (outputs must equal the head biases)."""
import configparser, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qbold_vi_amd.ops import Context, EncoderWeights
from qbold_vi_amd.init import init_encoder_weights
cfg = configparser.ConfigParser(); cfg.read(os.path.join(ROOT, "config")); p = dict(cfg["DEFAULT"])
p.update(tau_start="-0.015", tau_end="0.065", tau_step="0.00125")
ctx = Context(p, True, True)
T, L = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = init_encoder_weights(T=T, U=256, L=L, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=3)
for k in w:
    if k != "meta" and hasattr(w[k], "shape"):
        w[k] = np.zeros_like(w[k])
w["bf"] = np.arange(10, 15, dtype=np.float32)
w["bs"] = (100 + np.arange(T)).astype(np.float32) * 0.01
ew = EncoderWeights(ctx, T, 256, L, True, 0.0).set_from_arrays(w)
n = 200
x = torch.rand((n, T), device="cuda") * 0.5 + 0.2
_, q, s = ctx.encoder_fwd(ew, x, want=("out2", "sigma"))
torch.cuda.synchronize()
print("q[0]", q[0].cpu().numpy(), " q[131]", q[131].cpu().numpy())
ls = torch.log(s).cpu().numpy()
print("ls[0][:12]", np.round(ls[0][:12], 3)); print("ls[0][28:40]", np.round(ls[0][28:40], 3)); print("ls[199][52:]", np.round(ls[199][52:], 3))
print("rows equal across voxels:", bool(np.allclose(ls, ls[0:1], atol=1e-5)), bool(torch.allclose(q, q[0:1])))

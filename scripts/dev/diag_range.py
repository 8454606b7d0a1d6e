import configparser, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, init_weights, synth_inputs
from qbold_vi_amd.ops import Context, EncoderWeights
cfg = configparser.ConfigParser(); cfg.read(os.path.join(ROOT, "config")); params = dict(cfg["DEFAULT"])
o = Oracle("f32", params); ctx = Context(params, True, True)
w = init_weights(T=11, U=60, L=2, seed=3)
rng = np.random.default_rng(7)
for n in ("b0", "bc", "br1", "br2", "bg", "bf"):
    w[n] = (rng.standard_normal(w[n].shape) * 0.1).astype(np.float32)
w["gate_offset"] = -3.0
w2 = dict(w); w2["W0"] = w["W0"] * 3e5; w2["b0"] = w["b0"] * 3e5; w2["Wf"] = w["Wf"] / 3e5; w2["Ws"] = w["Ws"] / 3e5
w2["Wr1"] = w["Wr1"] * 0; w2["br1"] = w["br1"] * 0
ew = EncoderWeights(ctx, 11, 60, 2, True, -3.0).set_from_arrays(w2)
x, _ = synth_inputs(512, seed=12, oracle=o)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
prior, q_want, sigma = o.encoder_fwd(w2, x)
sums, q, nk = ctx.vi_fwd(ew, dev(x), dev(np.ones(512, np.float32)), dev(prior), 4, 8, seed=3)
print("sums", sums.cpu().numpy(), "nk nan count", int(torch.isnan(nk).sum()), "q nan", int(torch.isnan(q).sum()))
print("q err", np.abs(q.cpu().numpy() - q_want).max(), "q_want range", np.abs(q_want).max())
o1, o2, sg = ctx.encoder_fwd(ew, dev(x))
print("encoder_fwd nan counts", int(torch.isnan(o1).sum()), int(torch.isnan(o2).sum()), int(torch.isnan(sg).sum()))

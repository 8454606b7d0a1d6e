"""Dev timing of the headline step (fused qbold_vi_fwd, 1 M voxels x 11 tau, S = 32, K = 70) under kernel selections,
interleaved in one process on one box: python scripts/dev/time_headline.py [sel ...]  (default: 0 8)."""
import configparser, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from qbold_vi_amd.init import init_encoder_weights
from qbold_vi_amd.ops import EncoderWeights
cfg = configparser.ConfigParser(); cfg.read(os.path.join(ROOT, "config")); p = dict(cfg["DEFAULT"])
sels = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()] or [0, 8]
prec = "bf16" if "bf16" in sys.argv else "f32"
T = 24 if "p24" in sys.argv else 11
if T == 24:
    p.update(tau_start="-0.028", tau_end="0.065", tau_step="0.004")
n = 1 << 20
ctx, x = bench.make_inputs(n, p, seed=1, device=torch.device("cuda:0"))
w = init_encoder_weights(T=T, U=60, L=2, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=1)
ew = EncoderWeights(ctx, T, 60, 2, True, -3.0, precision=prec).set_from_arrays(w)
mask = torch.ones(n, device="cuda")
prior = ctx.encoder_fwd(ew, x, want=("out1",))[0]
out = (torch.empty(3, dtype=torch.float64, device="cuda"), torch.empty((n, 5), device="cuda"), torch.empty((n, 2), device="cuda"))
def run(k):
    for _ in range(k):
        ctx.vi_fwd(ew, x, mask, prior, 32, 70, seed=1, out=out)
t0 = time.time()
while time.time() - t0 < 0.3:
    run(20); torch.cuda.synchronize()
for rep in range(3):
    for sel in sels:
        ctx.set_kernel_selection(sel)
        run(20); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(100); b.record(); torch.cuda.synchronize()
        s = out[0].cpu()
        print(f"T={T} {prec} sel {sel:3d}: {a.elapsed_time(b) / 100:.4f} ms   -ELBO {float((s[0] + s[1]) / s[2]):.6f}", flush=True)

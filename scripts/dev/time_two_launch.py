#!/usr/bin/env python3
"""Times the headline workload as ONE launch (vi_fwd) and as TWO (encoder_fwd, then elbo_fwd on its q / sigma)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from qbold_vi_amd.init import init_encoder_weights
from qbold_vi_amd.ops import Context, EncoderWeights
from qbold_vi_amd.training import get_params
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
ctx = Context(get_params(os.path.join(root, "config")), True, True)
w = init_encoder_weights(T=ctx.T, U=60, L=2, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=1)
ew = EncoderWeights(ctx, ctx.T, 60, 2, True, -3.0).set_from_arrays(w)
N, S, K = 1 << 20, 32, 70
y = torch.stack([torch.rand(N, device="cuda") * 0.7 + 0.08, torch.rand(N, device="cuda") * 0.1 + 0.005], -1)
x = ctx.signal_fwd(y)
prior = ctx.encoder_fwd(ew, x, want=("out1",))[0]
mask = torch.ones(N, device="cuda")

def one():
    return ctx.vi_fwd(ew, x, mask, prior, S, K, seed=3, want_q=False, per_voxel=False)[0]

def two():
    _, q2, sg = ctx.encoder_fwd(ew, x, want=("out2", "sigma"))
    return ctx.elbo_fwd(x, mask, q2, prior, sg, S, K, seed=3, per_voxel=False)[0]

def timed(fn, steps=100):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps

a, b = one(), two()
print("neg_elbo one launch", float((a[0] + a[1]) / a[2]), "two launches", float((b[0] + b[1]) / b[2]))
for _ in range(2):
    print("one launch %.4f ms   two launches %.4f ms" % (timed(one), timed(two)))

#!/usr/bin/env python3
"""Times every data-path entry point of the C ABI on 1 M voxels (HIP events on the launch stream) and
prints one JSON object: ms per call and the algorithmic GB/s each call moves.  The headline number is
bench.py's; this table backs the per-kernel rows of DESIGN.md."""
import configparser
import json
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from qbold_vi_amd.init import init_encoder_weights  # noqa: E402
from qbold_vi_amd.ops import Context, EncoderWeights, TrainState  # noqa: E402


def main():
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "config"))
    params = dict(cfg["DEFAULT"])
    ctx = Context(params, True, True)
    T, n = ctx.T, 1 << 20
    w = init_encoder_weights(T=T, U=60, L=2, channelwise_gating=True, resid_init_std=0.05, seed=1, spatial_taps=9)
    ew = EncoderWeights(ctx, T, 60, 2, True, -3.0, spatial_taps=9).set_from_arrays(w)
    st = TrainState(ctx, ew)
    g = torch.Generator(device="cuda").manual_seed(0)
    y = torch.stack([torch.rand(n, generator=g, device="cuda") * 0.7 + 0.08,
                     torch.rand(n, generator=g, device="cuda") * 0.1 + 0.005], -1)
    x = ctx.signal_fwd(y)
    o1, q, sg = ctx.encoder_fwd(ew, x)
    ls = torch.log(sg)
    mask = torch.ones(n, device="cuda")
    z = ctx.normals(n, 1).reshape(n, 2)
    gs = torch.randn(n, T, device="cuda")
    y3 = torch.cat([y, y[:, :1]], -1).contiguous()
    B, X, Y, Z = 64, 32, 32, 16   # 1 M voxels as crops
    q5, m5 = q.reshape(B, X, Y, Z, 5), mask.reshape(B, X, Y, Z)
    x5 = x.reshape(B, X, Y, Z, T)
    gq0 = torch.zeros(n, 5, device="cuda")

    def fwd_train():
        st.forward(x, 2)

    def bwd_train():
        st.backward(2, gq0, gs, None)

    st.forward(x, 2)
    calls = {
        # name: (callable, algorithmic bytes per voxel)
        "signal_fwd": (lambda: ctx.signal_fwd(y), 8 + 4 * T),
        "signal_bwd": (lambda: ctx.signal_bwd(y, gs), 8 + 4 * T + 8),
        "signal_fwd_ex(hct)": (lambda: ctx.signal_fwd_ex(y, mask * 0.34), 12 + 4 * T),
        "normalise": (lambda: ctx.normalise(x), 8 * T),
        "encoder_fwd(stream 1)": (lambda: ctx.encoder_fwd(ew, x, want=("out1",)), 4 * T + 20),
        "encoder_fwd(stream 2 + sigma)": (lambda: ctx.encoder_fwd(ew, x, want=("out2", "sigma")), 8 * T + 20),
        "reparam": (lambda: ctx.reparam(q, z), 20 + 8 + 8),
        "logit_mvn_nlogp": (lambda: ctx.logit_mvn_nlogp(y, q), 8 + 20 + 4),
        "posterior_moments(20 draws)": (lambda: ctx.posterior_moments(q, 20, seed=1), 20 + 24),
        "posterior_moments(200 draws)": (lambda: ctx.posterior_moments(q, 200, seed=1), 20 + 24),
        "kl_fwd(K=70)": (lambda: ctx.kl_fwd(q, o1, K=70, seed=1), 44),
        "kl_closed": (lambda: ctx.kl_closed(q, o1), 44),
        "kl_diag(+grad)": (lambda: ctx.kl_diag(q, o1, mask, g_q=gq0), 44 + 40),
        "elbo_fwd(S=32,K=70)": (lambda: ctx.elbo_fwd(x, mask, q, o1, sg, 32, 70, seed=1), 8 * T + 52),
        "elbo_bwd(S=1,K=70)": (lambda: ctx.elbo_bwd(x, mask, q, o1, ls, 1, 70, seed=1), 12 * T + 72),
        "vi_fwd(S=32,K=70)": (lambda: ctx.vi_fwd(ew, x, mask, o1, 32, 70, seed=1), 4 * T + 52),
        "synth_loss_bwd": (lambda: st.synth_loss_bwd(y3, o1), 12 + 20 + 24),
        "wls_fit": (lambda: ctx.wls_fit(x), 4 * T + 12),
        "smoothness(+grad)": (lambda: ctx.smoothness(q5, m5, weight=5.0, g_q=gq0), 24 + 40),
        "encoder_train_fwd(stream 2)": (fwd_train, None),
        "encoder_train_bwd(stream 2)": (bwd_train, None),
        "encoder_spatial_fwd(3x3x1)": (lambda: st.forward_spatial(x5), None),
        "adamw_step(146k params)": (lambda: st.adamw(1e-3, 1e-4), None),
    }
    out = {}
    for name, (fn, bpv) in calls.items():
        import time
        t0 = time.perf_counter()   # clock ramp: ~20 ms of load before an idle MI355X runs at sustained clocks
        while time.perf_counter() - t0 < 0.1:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": round(ms, 4)}
        if bpv:
            out[name]["algorithmic_GBps"] = round(bpv * n / ms / 1e6, 1)
    print(json.dumps({"voxels": n, "T": T, "calls": out}, indent=1))


if __name__ == "__main__":
    main()

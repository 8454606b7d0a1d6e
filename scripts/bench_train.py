#!/usr/bin/env python3
"""Times one fine-tuning step (encoder forward with saved activations, ELBO backward, encoder
backward, AdamW) on a voxel batch and on a crop batch.  Not the headline bench (bench.py); used
with `rocprofv3 --kernel-trace --stats` to see where a training step goes."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from qbold_vi_amd.init import init_encoder_weights  # noqa: E402
from qbold_vi_amd.ops import Context, EncoderWeights, TrainState  # noqa: E402
from qbold_vi_amd.training import get_params  # noqa: E402


def measure(voxels=1 << 20, crops=(38, 25, 25, 8), steps=10, S=1, K=70, graph=True, only=None, ksel=0,
            config_dir="config"):
    """ms per fine-tuning step on a voxel batch and on a crop batch (bench.py embeds this in its JSON line)."""
    a = argparse.Namespace(voxels=voxels, crops=list(crops), steps=steps, S=S, K=K, graph=graph, only=only, ksel=ksel)
    return _run(a, config_dir)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=1 << 20)
    ap.add_argument("--crops", type=int, nargs=4, default=[38, 25, 25, 8], metavar=("B", "X", "Y", "Z"))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--S", type=int, default=1)
    ap.add_argument("--K", type=int, default=70)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--only", choices=["voxel", "crop"], default=None)
    ap.add_argument("--ksel", type=int, default=0, help="qbold_ctx_set_kernel_selection mask (QBOLD_KSEL_* of include/qbold_hip.h)")
    a = ap.parse_args()
    print(json.dumps(_run(a, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "config"))))


def _run(a, config_dir):
    params = get_params(config_dir)
    ctx = Context(params, True, True)
    if a.ksel:
        ctx.set_kernel_selection(a.ksel)
    w = init_encoder_weights(T=ctx.T, U=60, L=2, channelwise_gating=True, resid_init_std=0.05,
                             im_loss_sigma=0.05, seed=1, spatial_taps=9)
    ew = EncoderWeights(ctx, ctx.T, 60, 2, True, -3.0, spatial_taps=9).set_from_arrays(w)
    st = TrainState(ctx, ew)
    out = {}

    def timed(fn):
        import time
        t0 = time.perf_counter()   # clock ramp: an idle MI355X needs ~20 ms of load to reach sustained clocks
        while time.perf_counter() - t0 < 0.15:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.steps

    N = a.voxels
    y = torch.stack([torch.rand(N, device="cuda") * 0.7 + 0.08, torch.rand(N, device="cuda") * 0.1 + 0.005], -1)
    x = ctx.signal_fwd(y)
    prior = ctx.encoder_fwd(ew, x, want=("out1",))[0]
    mask = torch.ones(N, device="cuda")

    def voxel_step():
        q, ls = st.forward(x, 2)
        sums, gq, gls, _ = ctx.elbo_bwd(x, mask, q, prior, ls, a.S, a.K, seed=st.step)
        st.backward(2, gq, gls, sums)
        st.adamw(5e-3, 2e-4, 0.9, 0.9, 1e-7)

    if a.only != "crop":
        ms = timed(voxel_step)
        out["voxel_batch"] = dict(voxels=N, ms_per_step=ms, voxels_per_s=N / ms * 1e3)

    B, X, Y, Z = a.crops
    V = B * X * Y * Z
    x5 = x[:V].reshape(B, X, Y, Z, ctx.T).contiguous()
    m5 = torch.ones(B, X, Y, Z, device="cuda")
    p5 = prior[:V].contiguous()

    def crop_step():
        q, ls = st.forward_spatial(x5)
        sums, gq, gls, _ = ctx.elbo_bwd(x5.reshape(V, -1), m5.reshape(V), q, p5, ls, a.S, a.K, seed=st.step)
        ctx.smoothness(q.reshape(B, X, Y, Z, 5), m5, weight=5.0, g_q=gq)
        st.backward_spatial(gq, gls, sums)
        st.adamw(5e-3, 2e-4, 0.9, 0.9, 1e-7)

    if a.only == "voxel":
        return out
    ms = timed(crop_step)
    out["crop_batch"] = dict(crops=[B, X, Y, Z], voxels=V, ms_per_step=ms, voxels_per_s=V / ms * 1e3)
    if a.graph:
        # How much of the step is launch gaps: the same ~50 launches captured into a hipGraph (the C ABI neither
        # allocates nor synchronises, so stream capture works through it) and replayed.  A replay issues the kernels
        # back to back from the device's own queue, so replay time ~ the sum of the kernels' times, and eager - replay
        # is what the host-side launches add.  The kernel-by-kernel sum itself: profiles/rNN_train_step_kernel_stats.csv.
        try:
            sidestream = torch.cuda.Stream()
            with torch.cuda.stream(sidestream):
                crop_step()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=sidestream):
                    crop_step()
                for _ in range(2):
                    g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.steps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
            rep = e0.elapsed_time(e1) / a.steps
            out["crop_batch"].update(graph_replay_ms=rep, launch_gap_ms=ms - rep)
        except Exception as e:   # a diagnostic: never takes the timing down
            out["crop_batch"]["graph_replay_error"] = repr(e)
    return out


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- voxel-ELBO evaluations per second on MI355X (BASELINE.json metric).

A step = one pass of the whole hot path (qbold_vi_fwd: encoder stream 2 -> S reparameterised
draws -> forward model over T taus -> NLL + K-draw MC KL -> masked sums) over one batch of
synthetic voxels already resident in HBM.  Workload at N=1: BASELINE.json configs[1]
(1 M voxels x 11 tau, S=32, K=70, fp32, configurations/optimal.yaml encoder).  With N>1 ranks
(one process per GPU, torchrun) every rank owns its own 1 M-voxel shard (weak scaling) and the three
masked sums are all-reduced over RCCL every step.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel vi_fwd_kernel, timed
with HIP events on the launch stream) and "cpu_baseline" (the oracle's literal-Simpson restatement
on the host cores, bounded sample).
"""
import argparse
import configparser
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def measured_profile(name, kernel):
    """A committed PMC summary under profiles/ (written by scripts/collect_profiles.py from rocprofv3 --pmc runs
    of this same command) -- used only if it was measured on the kernel sources of this build; stale files
    are dropped, not quoted."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        d["_file"] = name
        from qbold_vi_amd.build import source_fingerprint
        if d.get("source_sha256") != source_fingerprint(d.get("source_units")) or kernel not in d.get("kernel", ""):
            return None
        return d
    except Exception:
        return None


F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / f16 MFMA peak
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_voxel(T):
    # read signal 4T + mask 4 + prior 20; write q 20 + (nll, kl) 8
    return 4 * T + 4 + 20 + 20 + 8


def encoder_macs_per_voxel(T, U, L):
    # stream 2 only: first layer, L x (skip + 2 residual + gate), q head, sigma head
    return T * U + L * 4 * U * U + U * 5 + U * T


def algorithmic_flops_per_voxel(T, U, L, S, K):
    # encoder 2 flop/MAC; forward model ~60 flop and NLL ~8 flop per (draw, tau);
    # KL ~80 flop per draw.  = 90,376 at optimal.yaml, S=32, K=70.
    return 2 * encoder_macs_per_voxel(T, U, L) + S * (60 * T + 8 * T) + 80 * K


def make_inputs(n, params, seed, device):
    """SURVEY 8(d) synthetic voxels, generated ON the GPU by the library's own forward model
    (the oracle is not involved): OEF = clip(N(0.4,0.2),.05,.8), DBV = TruncNormal(0.025,0.02;
    [.003,.195]) by rejection, reference noise model for T=11 (signals.py:116-128)."""
    import torch
    from qbold_vi_amd.ops import Context
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    oef = (torch.randn(n, generator=g, device=device) * 0.2 + 0.4).clamp_(0.05, 0.8)
    dbv = torch.randn(n, generator=g, device=device) * 0.02 + 0.025
    for _ in range(64):
        bad = (dbv < 0.003) | (dbv > 0.195)
        if not bool(bad.any()):
            break
        dbv = torch.where(bad, torch.randn(n, generator=g, device=device) * 0.02 + 0.025, dbv)
    dbv.clamp_(0.003, 0.195)
    ctx = Context(params, True, True, device=device)
    sig = ctx.signal_fwd(torch.stack([oef, dbv], -1))
    if ctx.T == 11:
        norm_snr = torch.tensor([0.985, 1.00, 1.01, 1., 0.97, 0.95, 0.93, 0.90, 0.86, 0.83, 0.79],
                                device=device)
    else:  # the reference defines the noise model for 11 / 24 taus only: flat SNR
        norm_snr = torch.ones(ctx.T, device=device)
    snr = (torch.rand(n, 1, generator=g, device=device) * 70 + 50) * norm_snr[None]
    std = sig.mean(0, keepdim=True) / snr
    sig = sig + torch.randn(sig.shape, generator=g, device=device) * std
    return ctx, sig.contiguous()


def cpu_baseline(params, weights_np, S, K, budget_s=15.0):
    """The reference algorithm on the host cores (BASELINE: "the reference's own CPU path timed beside it"; the
    reference's TensorFlow cannot run here, so these are the two restatements of oracle/):
      value / cases  -- oracle/qbold_oracle.c (C + OpenMP: literal 129-node Simpson + Cephes j0f per (voxel,
                        sample, tau), encoder, logit-Normal KL) on bounded samples: SURVEY 8(d)'s three cases
                        (i) forward model only, (ii) ELBO at the reference defaults S = 1 / K = 70, (iii) the
                        bench workload's S;
      op_granularity -- oracle/torch_ref.py: the same arithmetic as whole-batch torch float32 ops with the
                        reference's materialised [V, T, 129] tensor and S-fold tiled batch, chunked like
                        signals.py:281-285 -- how the reference itself runs on a CPU.
    Baselines, not targets: a port in C is faster than the reference's op-by-op execution."""
    from oracle.oracle import Oracle, synth_inputs
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    orc = Oracle("f32", params, threads=cores)
    T = orc.T
    weights_np = dict(weights_np, gate_offset=-3.0, meta=dict(T=T, U=int(np.asarray(weights_np["W0"]).shape[1]),
                                                               L=int(np.asarray(weights_np["Wc"]).shape[0]),
                                                               channelwise_gating=True))

    def run_elbo(n, s):
        x, _ = synth_inputs(n, params, seed=2, oracle=orc)
        t0 = time.perf_counter()
        prior, q, sigma = orc.encoder_fwd(weights_np, x)
        zs = orc.philox_normals(1, 0, 0, n, s)
        zk = orc.philox_normals(1, 1, 0, n, K)
        orc.elbo(x, np.ones(n, np.float32), q, prior, sigma, zs, zk)
        return time.perf_counter() - t0

    def run_fwd(n):
        _, y = synth_inputs(16, params, seed=2, oracle=orc)
        y = np.tile(y, (n // 16 + 1, 1))[:n]
        t0 = time.perf_counter()
        orc.signal_fwd(y)
        return time.perf_counter() - t0

    def bounded(fn, n0, budget, *a):
        t_probe = fn(n0, *a)
        n = int(min(max(n0, n0 * budget / max(t_probe, 1e-3)), 1 << 20))
        n = max(1024, (n // 1024) * 1024)
        return n, fn(n, *a)

    n3, t3 = bounded(run_elbo, 256 * cores, 0.45 * budget_s, S)
    n2, t2 = bounded(run_elbo, 256 * cores, 0.15 * budget_s, 1)
    n1, t1 = bounded(run_fwd, 1024 * cores, 0.10 * budget_s)
    out = {"value": n3 / t3, "unit": "voxel-ELBO evals/s", "cores": cores, "kind": "port",
           "sample": f"{n3} voxels x {T} tau, S={S}, K={K}, C oracle (OpenMP), literal Simpson-129 "
                     f"+ Cephes j0f as the reference computes it; {t3:.1f} s",
           "cases": {"i_forward_model_only": {"value": n1 / t1, "unit": "(OEF, DBV) -> signal evals/s",
                                              "sample": f"{n1} pairs x {T} tau; {t1:.2f} s"},
                     "ii_elbo_reference_defaults_S1_K70": {"value": n2 / t2, "unit": "voxel-ELBO evals/s",
                                                           "sample": f"{n2} voxels, S=1, K={K}; {t2:.2f} s"},
                     "iii_elbo_bench_workload": {"value": n3 / t3, "unit": "voxel-ELBO evals/s",
                                                 "sample": f"{n3} voxels, S={S}, K={K}; {t3:.1f} s"}}}
    try:   # torch float32 at the reference's op granularity (second restatement; oracle/torch_ref.py)
        import torch
        from oracle import torch_ref as tr
        # intra-op threads: a GPU box gives one GPU's job a 16-core CPU share whatever the host's core count,
        # and torch's thread pool collapses when oversubscribed (256 threads: 100x slower)
        tthreads = max(1, min(cores, 16))
        torch.set_num_threads(tthreads)
        nt = 128
        x, _ = synth_inputs(4096, params, seed=2, oracle=orc)

        def run_torch(nv):
            xs = x[:nv]
            t0 = time.perf_counter()
            p1, q2, sg = tr.encoder(weights_np, xs, orc.se_idx, -3.0)
            zs, zk = orc.philox_normals(1, 0, 0, nv, S), orc.philox_normals(1, 1, 0, nv, K)
            tr.elbo(xs, np.ones(nv, np.float32), q2, p1, sg, zs, zk, params, orc.se_idx)
            return time.perf_counter() - t0
        tp = run_torch(nt)
        nv = int(min(4096, max(nt, nt * 0.3 * budget_s / max(tp, 1e-3)))) // 128 * 128
        tt = run_torch(nv) if (nv > nt and tp < 0.3 * budget_s) else tp
        nv = nv if (nv > nt and tp < 0.3 * budget_s) else nt
        out["op_granularity"] = {"value": max(nv, nt) / tt, "unit": "voxel-ELBO evals/s", "cores": tthreads, "kind": "port",
                                 "sample": f"{max(nv, nt)} voxels x {T} tau, S={S}, K={K}, torch float32 on the host: "
                                           f"[V, T, 129] Bessel tensor materialised in chunks of 4,096 rows, batch tiled "
                                           f"S-fold (signals.py:168-171, 281-285; model.py:245-246); {tt:.1f} s"}
    except Exception as e:  # the baseline is reported, never required
        out["op_granularity"] = {"error": repr(e)}
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks, argv):
    """`python3 bench.py --gpus N` without torchrun: start N rank processes (one per GPU, the launch
    contract's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment) and wait for them.  This parent
    never imports torch and never touches the GPU, nothing is exec'ed; the children inherit stdout, so
    rank 0's JSON line is this command's JSON line.  Any child failing ends the others (exact PIDs)
    and makes the exit code non-zero."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:      # a rank died: its peers would wait in the rendezvous forever
                    q.terminate()
    if rc != 0:
        print(f"bench.py: a rank exited with code {rc}", file=sys.stderr)
    return rc


def rehearse_launch(args, world, rank):
    """--rehearse_launch: the launcher, rendezvous, all-reduce ring, barrier-bracketed timing and the JSON
    line of the N-rank path with NO kernel in the step (host tensors, gloo) -- what a CPU-only container
    can check of `bench.py --gpus N`.  It reports no throughput ("value": null)."""
    import torch
    import torch.distributed as dist
    backend = os.environ.get("QBOLD_DIST_BACKEND", "gloo")
    if world > 1:
        dist.init_process_group(backend)
    RING = 4
    outs = [torch.zeros(3, dtype=torch.float64) for _ in range(RING)]
    pending = [None] * RING

    def step(k):
        slot = k % RING
        if pending[slot] is not None:
            pending[slot].wait()
            pending[slot] = None
        outs[slot][:] = torch.tensor([1.0 + rank, 2.0, 1.0], dtype=torch.float64)
        if world > 1:
            pending[slot] = dist.all_reduce(outs[slot], async_op=True)
        return outs[slot]

    def drain():
        for slot in range(RING):
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None

    for k in range(args.warmup):
        step(k)
    drain()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        sums = step(k)
    drain()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ms = torch.tensor([elapsed / max(args.steps, 1) * 1e3], dtype=torch.float64)
    all_ms = [torch.zeros_like(ms) for _ in range(world)]
    if world > 1:
        dist.all_gather(all_ms, ms)
    else:
        all_ms = [ms]
    all_ms = [float(t.item()) for t in all_ms]
    want = sum(1.0 + r for r in range(world))
    assert abs(float(sums[0]) - want) < 1e-12 and float(sums[2]) == world, "all-reduce of the sums is wrong"
    if rank == 0:
        print(json.dumps({"metric": "voxel-ELBO evals/sec", "value": None, "unit": "voxel-ELBO evals/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": max(all_ms), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "rehearsal": "launcher / rendezvous / all-reduce ring only: no kernel ran, no throughput",
                          "ranks_seen": dist.get_world_size() if world > 1 else 1,
                          "backend": dist.get_backend() if world > 1 else "none",
                          "rank_ms_per_step": {"min": min(all_ms), "max": max(all_ms)},
                          "config": {"workload": "none (launcher rehearsal)", "global_voxels": 0,
                                     "parallelism": f"voxel-shard x{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class Spec:
    """One workload of the bench: which BASELINE configuration, how many voxels, which encoder arithmetic."""

    def __init__(self, config=2, protocol=11, voxels=1 << 20, encoder_precision="f32", tissue="table", exact=False,
                 wide_posterior=False, off_grid=False):
        self.config, self.protocol, self.voxels = config, protocol, voxels
        self.encoder_precision, self.tissue, self.exact = encoder_precision, tissue, exact
        # the slow sides of the headline kernel's data-dependent switches:
        #   wide_posterior -- the posterior head's log-std biases at +3 (s = 3 tanh(3) - 1 = 1.99, e^s = 7.3): every
        #                     voxel's draws can reach the reference's logit clip, so every wave runs the GENERAL KL loop
        #                     (clipped logits, 19 instructions per draw) instead of the whitened form;
        #   off_grid       -- tau_start = -17 ms: image 2 still normalises (model.py:95) but its tau is -1 ms, so the
        #                     grid does not mirror about it: no table-free spin-echo signal, no shared tau pairs, no
        #                     per-tau table (the x-indexed table for all 11 taus).
        self.wide_posterior, self.off_grid = wide_posterior, off_grid


def run_workload(spec, args, steps, warmup, rank, device, use_pg, world, S, K):
    """Builds the workload's inputs in HBM, ramps the clocks, runs `warmup` untimed and `steps` timed steps
    (barrier + device synchronise on both sides), and returns the measurements of THIS rank plus what the
    roofline objects need.  One step = one pass of the hot path over the whole batch."""
    import ctypes as C

    import torch
    import torch.distributed as dist
    from qbold_vi_amd import _lib
    from qbold_vi_amd.init import init_encoder_weights
    from qbold_vi_amd.ops import EncoderWeights, _ptr, _stream

    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "config"))
    params = dict(cfg["DEFAULT"])
    T, U, L = 11, 60, 2  # configurations/optimal.yaml
    if spec.config == 3:  # 64 taus from -0.015 s in 1.25 ms steps (se_idx 12), width 256
        params.update(tau_start="-0.015", tau_end="0.065", tau_step="0.00125")
        T, U = 64, 256
    elif spec.protocol == 24:
        params.update(tau_start="-0.028", tau_end="0.065", tau_step="0.004")
        T = 24
    elif spec.off_grid:
        params.update(tau_start="-0.017", tau_end="0.071", tau_step="0.008")
    n = spec.voxels
    ctx, x = make_inputs(n, params, seed=1 + rank, device=device)
    ctx.set_tissue_mode(spec.tissue)
    w = init_encoder_weights(T=T, U=U, L=L, channelwise_gating=True, resid_init_std=0.05,
                             im_loss_sigma=0.05, seed=1)
    if spec.wide_posterior:
        w["bf"][1] = w["bf"][3] = 3.0
    if spec.encoder_precision == "bf16" and spec.config != 2:
        raise SystemExit("--encoder_precision bf16 applies to the fused kernel (config 2)")
    ew = EncoderWeights(ctx, T, U, L, True, -3.0, precision=spec.encoder_precision).set_from_arrays(w)
    mask = torch.ones(n, device=device)
    prior, _, _ = ctx.encoder_fwd(ew, x, want=("out1",))  # prior = stream-1 output (train.py:26-31)
    q_buf, nk_buf = torch.empty((n, 5), device=device), torch.empty((n, 2), device=device)
    # The three masked sums of every step are all-reduced (RCCL over xGMI).  A ring of result buffers
    # lets step k+1's kernel run while step k's 24-byte all-reduce is in flight on RCCL's stream: the
    # reduced ELBO is consumed a few steps later, as a training loop consumes its loss.
    RING = 4
    outs = [(torch.empty(3, dtype=torch.float64, device=device), q_buf, nk_buf) for _ in range(RING)]
    pending = [None] * RING
    voxel0 = rank * n

    # Config 3 runs as two launches (the one-launch wide encoder, then the ELBO kernel on its heads): the bench
    # issues them through the two C entry points qbold_vi_fwd itself chains, with an event between them, so that
    # the dominant kernel's duration is measured live in the timed region.
    two_launch = bool(getattr(ew, "fused_wide", False)) and not spec.exact
    if two_launch:
        ls_buf = torch.empty((n, T), device=device)
        ws = ctx._workspace()

    def launch(slot, mid_event=None):
        sums = outs[slot][0]
        if spec.exact:
            # the strictly-float32 encoder: layer-wise v_mfma_f32_16x16x4_f32 GEMMs (activations through HBM),
            # then the ELBO kernel on its heads -- the path vi_fwd(range_check=True) falls back to
            s2, _, _ = ctx.vi_fwd_exact(ew, x, mask, prior, S, K, seed=1, voxel0=voxel0)
            sums.copy_(s2)
            return sums
        if not two_launch:
            ctx.vi_fwd(ew, x, mask, prior, S, K, seed=1, voxel0=voxel0, out=outs[slot])
            return sums
        _lib.check(ctx.lib.qbold_encoder_fused_fwd(ctx.handle, C.byref(ew.shape), ew.fused_ptr(), _ptr(x), _ptr(q_buf),
                                                   _ptr(ls_buf), n, _stream()), "qbold_encoder_fused_fwd")
        if mid_event is not None:
            mid_event.record()
        _lib.check(ctx.lib.qbold_elbo_fwd_logsigma(ctx.handle, _ptr(x), _ptr(mask), _ptr(q_buf), _ptr(prior),
                                                   _ptr(ls_buf), int(S), int(K), 1, int(voxel0), _ptr(nk_buf),
                                                   _ptr(sums), _ptr(ws), n, _stream()), "qbold_elbo_fwd_logsigma")
        return sums

    def step(k, mid_event=None):
        slot = k % RING
        if pending[slot] is not None:
            pending[slot].wait()   # stream-level wait: the slot's previous all-reduce has finished
            pending[slot] = None
        sums = launch(slot, mid_event)
        if use_pg:
            pending[slot] = dist.all_reduce(sums, async_op=True)  # sum m*nll, sum kl, sum m
        return sums

    def drain():
        for slot in range(RING):
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp (setup, like input generation above): same work, never timed
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for k in range(8):
            step(k)
        drain()
        torch.cuda.synchronize()
    for k in range(warmup):
        step(k)
    drain()
    fence()
    # per-launch durations: HIP events on the launch stream
    every = max(1, min(args.event_every, steps // 5))   # at least five bracketed steps (short runs: every step)
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) if k % every == 0 else None
           for k in range(steps)]
    n_allreduce = 0
    t0 = time.perf_counter()
    for k, ev in enumerate(evs):
        slot = k % RING
        if pending[slot] is not None:
            pending[slot].wait()
            pending[slot] = None
        if ev is not None:
            ev[0].record()
        sums = launch(slot, ev[1] if (two_launch and ev is not None) else None)
        if ev is not None:
            ev[2].record()
        if use_pg:
            pending[slot] = dist.all_reduce(sums, async_op=True)
            n_allreduce += 1
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    evs = [ev for ev in evs if ev is not None]
    step_kernel_ms = float(np.mean([ea.elapsed_time(eb) for ea, em, eb in evs]))
    enc_kernel_ms = float(np.mean([ea.elapsed_time(em) for ea, em, eb in evs])) if two_launch else None
    s = sums.cpu().numpy()
    return dict(params=params, w=w, T=T, U=U, L=L, n=n, two_launch=two_launch, elapsed=elapsed,
                step_kernel_ms=step_kernel_ms, enc_kernel_ms=enc_kernel_ms, sums=s, n_allreduce=n_allreduce)


# Issue cost of one wave64 vector instruction on gfx950, cycles, measured with scripts/ubench/ubench*.hip (MEASUREMENTS.md 4.4):
# the class counters of rocprofv3 (SQ_INSTS_VALU_*) x these costs / SIMD-cycles = roofline.issue.model_frac
ISSUE_COST = {"SQ_INSTS_VALU_TRANS_F32": 8.45, "SQ_INSTS_VALU_CVT": 4.6, "SQ_INSTS_VALU_INT32": 4.0,
              "SQ_INSTS_VALU_FMA_F32": 2.9, "SQ_INSTS_VALU_MUL_F32": 2.9, "SQ_INSTS_VALU_ADD_F32": 2.9}
ISSUE_COST_OTHER = 3.5    # logic, moves, max / med3, compares, permlane: between the 2.9 and 4.4-cycle classes
N_SIMD = 256 * 4


def issue_object(prof):
    """How much of the chip's vector-issue time the kernel uses, from the counters of a profiles/rNN_*_pmc.json file
    (None without one).  One formula each:
      frac       = SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles          (SQ_ACTIVE_INST_* count quad-cycles)
      model_frac = sum_class(SQ_INSTS_VALU_class x ISSUE_COST[class]) / SIMD-cycles
      SIMD-cycles = GRBM_GUI_ACTIVE / 8 x 1024                      (the counter sums the 8 XCDs; 256 CUs x 4 SIMDs)
    frac near 1 = the kernel IS its vector instruction stream: the lever is the instruction count."""
    if not prof:
        return None
    c = prof.get("counters", {})
    if not all(k in c for k in ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU")):
        return None
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMD
    out = {"frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, "simd_cycles": simd_cycles,
           "valu_instructions": c["SQ_INSTS_VALU"],
           "cycles_per_valu_instruction": c["SQ_ACTIVE_INST_VALU"] * 4.0 / c["SQ_INSTS_VALU"],
           "formula": "SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024)",
           "source": "profiles/" + prof.get("_file", "")}
    if all(k in c for k in ISSUE_COST):
        classed = sum(c[k] for k in ISSUE_COST)
        cyc = sum(c[k] * v for k, v in ISSUE_COST.items()) + max(c["SQ_INSTS_VALU"] - classed, 0.0) * ISSUE_COST_OTHER
        out["model_frac"] = cyc / simd_cycles
        out["model_formula"] = ("(sum_k SQ_INSTS_VALU_k * cost_k + (SQ_INSTS_VALU - sum_k SQ_INSTS_VALU_k) * "
                                f"{ISSUE_COST_OTHER}) / SIMD-cycles, cost = " + json.dumps(ISSUE_COST))
        out["transcendental_share_of_instructions"] = c["SQ_INSTS_VALU_TRANS_F32"] / c["SQ_INSTS_VALU"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        out["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
    return out


def roofline_of(spec, m, S, K, step_kernel_ms, kernel_ms):
    """The roofline object of one workload.  Three figures, each from one formula:
      hbm      -- the roofline BASELINE.json's metric names: SURVEY 8(d)'s algorithmic bytes per voxel x voxels / the
                  step's kernel time, against the guide's 8 TB/s.  The path moves 96 B per voxel against ~90 kFLOP, so
                  this is ~0.03 by arithmetic whatever the kernel does.
      issue    -- what the counters say binds the fused kernel: vector-issue time used / available (issue_object).
      two_pipe -- SURVEY 8(d)'s flops priced at the guide's two dense peaks, serial sum, no overlap assumed: the
                  encoder's 2 x MACs at the f16 / bf16 MFMA peak (ONE pass -- the three split-f16 passes that
                  float32-grade products take are this implementation's cost, not required work), the rest at the f32
                  vector peak.  'survey_tflops' is SURVEY's flop count per second, not machine flops: SURVEY prices a
                  (draw, tau) evaluation at 60 flop, the kernel executes ~12 instructions for it.
    Top level (the contract's keys): config 2 -> the hbm figure (bound "hbm"); config 3 -> its dominant kernel, the
    one-launch encoder, against the dense f16 MFMA peak (bound "mfma")."""
    T, U, L, n, two_launch = m["T"], m["U"], m["L"], m["n"], m["two_launch"]
    enc_flops_v = 2.0 * encoder_macs_per_voxel(T, U, L)
    flops_v = float(algorithmic_flops_per_voxel(T, U, L, S, K))
    passes = 1.0 if spec.encoder_precision == "bf16" else 3.0
    bytes_v = algorithmic_bytes_per_voxel(T)
    if spec.exact:   # every flop of this path, the encoder's included, runs at the f32 rate
        t_min = flops_v / (F32_MFMA_PEAK_TFLOPS * 1e12) * n
    else:
        t_min = (enc_flops_v / (BF16_MFMA_PEAK_TFLOPS * 1e12) + (flops_v - enc_flops_v) / (F32_MFMA_PEAK_TFLOPS * 1e12)) * n
    survey_tflops = flops_v * n / (step_kernel_ms * 1e-3) / 1e12
    two_pipe = {"frac": t_min / (step_kernel_ms * 1e-3), "floor_ms": t_min * 1e3, "kernel_ms": step_kernel_ms,
                "survey_tflops": survey_tflops, "survey_flops_per_voxel": flops_v,
                "encoder_flops_per_voxel": enc_flops_v, "encoder_mfma_passes_executed": passes,
                "peaks_tflops": {"f16_bf16_mfma_dense": BF16_MFMA_PEAK_TFLOPS, "f32_vector": F32_MFMA_PEAK_TFLOPS},
                "formula": ("(encoder_flops / f16_mfma_peak + (survey_flops - encoder_flops) / f32_vector_peak) * voxels "
                            "/ kernel time" if not spec.exact else "survey_flops / f32_vector_peak * voxels / kernel time")}
    hbm_gbs = bytes_v * n / (step_kernel_ms * 1e-3) / 1e9
    hbm = {"algorithmic_bytes_per_voxel": bytes_v, "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": hbm_gbs / HBM_PEAK_GBS, "formula": "algorithmic_bytes_per_voxel * voxels / kernel time / 8 TB/s"}
    prof = None
    if two_launch:
        kname = "wide_fused_kernel<4, 2, true>"
        prof = measured_profile(PROFILE_TAG + "_config3_pmc.json", "wide_fused_kernel") if n == 1 << 20 else None
        ach = enc_flops_v * n / (kernel_ms * 1e-3) / 1e12
        top = {"bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": ach / BF16_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_voxel": enc_flops_v,
               "three_pass": {"peak": BF16_MFMA_PEAK_TFLOPS / passes, "frac": ach * passes / BF16_MFMA_PEAK_TFLOPS,
                              "note": "the same launch against peak / 3: float32-grade products take three f16 MFMA "
                                      "passes (hi.hi, hi.lo, lo.hi) in this implementation"}}
        note = ("dominant kernel = the one-launch wide encoder (activations in registers, weights streamed L2 -> LDS): "
                "'achieved' = SURVEY 8(d)'s encoder flops per voxel x voxels / its launch duration (HIP events inside "
                "the timed region) against the guide's dense f16 MFMA peak; under this kernel's matrix load the chip "
                "holds ~1.6-2.0 GHz, not the 2.4 GHz the peak assumes (MEASUREMENTS.md 4.7); the second launch "
                "(elbo_fwd_gt64_kernel, vector-issue-bound) is in 'two_pipe' / 'hbm', which cover the whole step")
    else:
        if spec.exact:
            kname = "xw64_kernel / xw64_fork_kernel / xw64_gate_kernel / xw64_heads_kernel + elbo_fwd_kernel"
            note = ("the strictly-float32 encoder (v_mfma_f32_16x16x4_f32 GEMMs, one launch per layer, float32 activations "
                    "through HBM) + the ELBO kernel; the layer GEMMs are HBM-bound on their activation traffic "
                    "(MEASUREMENTS.md 4.5), which the algorithmic-byte figure here does not count")
        else:
            kname = "vi_fwd_kernel"
            pmc_name = {(11, "f32"): "_vi_fwd_pmc.json", (24, "f32"): "_p24_vi_fwd_pmc.json",
                        (11, "bf16"): "_bf16_vi_fwd_pmc.json"}.get((spec.protocol, spec.encoder_precision))
            prof = measured_profile(PROFILE_TAG + pmc_name, "vi_fwd_kernel") \
                if (pmc_name and spec.config == 2 and spec.tissue == "table" and n == 1 << 20
                    and not spec.wide_posterior and not spec.off_grid) else None
            note = ("one launch; top level = the HBM roofline the metric names (algorithmic bytes / kernel time against "
                    "8 TB/s): ~0.03 by arithmetic, the path is compute-bound by two orders of magnitude (SURVEY H1). "
                    "What binds the kernel is in 'issue' (vector-issue time used / available, from the counters) and the "
                    "flop-based composite in 'two_pipe'; quote the three together")
        top = {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
               "algorithmic_bytes_per_voxel": bytes_v}
    counters, traffic = {}, None
    if prof:
        traffic = prof.get("hbm_bytes_per_launch")
        counters = dict(prof.get("counters", {}), source="profiles/" + prof.get("_file", ""),
                        measured_on_sources=prof.get("source_sha256", "")[:12], kernel=prof.get("kernel"))
    issue = issue_object(prof)
    return {"kernel": kname, **top, "traffic": traffic, "kernel_ms": kernel_ms,
            "limiter": ("matrix pipe (power-limited clock)" if two_launch else "HBM (layer-wise activations)" if spec.exact
                        else "vector issue"),
            "hbm": hbm, "issue": issue, "two_pipe": two_pipe,
            **({"counters": counters} if counters else {}),
            "step": {"launches": (["wide_fused_kernel", "elbo_fwd_gt64_kernel", "reduce_partials_kernel"] if two_launch
                                  else ["layer-wise encoder (9 launches)", "elbo_fwd_kernel", "reduce_partials_kernel"]
                                  if spec.exact else ["vi_fwd_kernel", "reduce_partials_kernel"]),
                     "kernel_ms": step_kernel_ms,
                     **({"traffic": prof["step"].get("hbm_bytes"), "algorithmic_bytes": prof["step"].get("algorithmic_bytes")}
                        if (prof and "step" in prof) else {})},
            "note": note}


def workload_text(spec, m, S, K):
    T, U, L, n = m["T"], m["U"], m["L"], m["n"]
    if spec.exact:
        arith = "exact float32 (v_mfma_f32_16x16x4_f32, one launch per layer)"
    elif spec.encoder_precision == "f32":
        arith = ("operands split in two f16 halves, three MFMA passes (hi.hi, hi.lo, lo.hi), f32 accumulate: "
                 "float32-grade (~22 significant bits), |x| < 65504")
    else:
        arith = "operands rounded to bf16, one MFMA pass, f32 accumulate"
    entry = ("qbold_encoder_train_fwd + qbold_elbo_fwd (vi_fwd_exact)" if spec.exact else
             "qbold_encoder_fused_fwd + qbold_elbo_fwd_logsigma (= qbold_vi_fwd)" if m["two_launch"] else "fused qbold_vi_fwd")
    slow = ("; posterior log-std biases +3: every wave on the general (clipped-logit) KL loop" if spec.wide_posterior else
            "; tau_start -17 ms: spin-echo image off tau = 0, no mirrored pairs, x-indexed table" if spec.off_grid else "")
    return (f"{n} synthetic voxels/GPU x {T} tau, S={S} likelihood draws, K={K} KL draws, encoder U={U}, L={L} "
            f"({'optimal.yaml' if U == 60 else 'BASELINE config 3'}), {entry}, tissue integral: {spec.tissue}; "
            f"encoder arithmetic: {arith}; sampling, forward model and ELBO sums: f32{slow}")


PROFILE_TAG = "r04"
# what the default N = 1 run times after the headline: every configuration a summary
# quotes, so that each number has a driver-witnessed line
VARIANTS = (("config3_64tau_width256", dict(config=3)),
            ("bf16_encoder", dict(encoder_precision="bf16")),
            ("voxels_4194304", dict(voxels=4194304)),
            ("protocol_24tau", dict(protocol=24)),
            ("exact_f32_encoder", dict(exact=True)),
            ("general_kl_loop_wide_posteriors", dict(wide_posterior=True)),
            ("off_grid_spin_echo", dict(off_grid=True)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp_ms", type=float, default=150.0,
                    help="untimed steps run for this long BEFORE the warm-up steps: an idle MI355X needs ~20 ms of "
                         "load to reach its sustained clocks (0.70 ms/step right after idle, 0.61 ms sustained)")
    ap.add_argument("--voxels", type=int, default=1 << 20, help="voxels per GPU")
    ap.add_argument("--mc_samples", type=int, default=32)
    ap.add_argument("--kl_samples", type=int, default=70)
    ap.add_argument("--tissue", choices=["table", "literal"], default="table")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="BASELINE.json config: 2 = 11 tau / width 60 (headline), 3 = 64 tau / width 256")
    ap.add_argument("--protocol", type=int, default=11, choices=[11, 24],
                    help="config 2 only: 11 = tau -16..64 ms in 8 ms steps (headline, the config file); 24 = the "
                         "reference's second protocol, tau -28..64 ms in 4 ms steps (signals.py:120-121)")
    ap.add_argument("--encoder_precision", choices=["f32", "bf16"], default="f32",
                    help="f32 (headline): float32-grade split-f16 MFMA; bf16: BASELINE config 5's "
                         "'bf16 forward / fp32 ELBO accum' (encoder products on bf16 operands)")
    ap.add_argument("--exact_f32", action="store_true",
                    help="the strictly-float32 encoder (f32-input MFMA, layer-wise) instead of the split-f16 one")
    ap.add_argument("--rehearse_launch", action="store_true",
                    help="run the N-rank launcher, rendezvous, all-reduce ring and JSON line with no kernel in the step "
                         "(host tensors over gloo; reports no throughput) -- the CPU-container check of --gpus N")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_variants", action="store_true",
                    help="skip the variant workloads the default N = 1 headline run times after the headline")
    ap.add_argument("--variants", action="store_true", help="time the variant workloads even on a non-default run")
    ap.add_argument("--variant_steps", type=int, default=20)
    ap.add_argument("--event_every", type=int, default=8,
                    help="HIP events bracket every M-th timed step (a timing event is a barrier packet in the queue: "
                         "around every step they cost up to 5 %% of a 0.5 ms step on some boxes)")
    ap.add_argument("--cpu_budget_s", type=float, default=15.0)
    args = ap.parse_args()
    if os.environ.get("QBOLD_DEBUG_SKIP", "0") not in ("", "0"):
        # The default build of the library has no ablation hook at all (they compile in only with -DQBOLD_ABLATION,
        # scripts/dev/build_ablation.sh); with such a build loaded, the run is marked so that the phases switched
        # off for timing experiments (MEASUREMENTS.md 4.4 / 4.7) never reach a reported number.
        os.environ["QBOLD_ALLOW_ABLATION"] = "1"
        print("bench.py: QBOLD_DEBUG_SKIP is set -- an ablation build of the library would skip work; this run is an "
              "ablation, not a benchmark", file=sys.stderr)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: become the launcher BEFORE anything touches the GPU (torch is not even imported)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    if args.rehearse_launch:
        return rehearse_launch(args, world, rank)

    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: no ROCm device is visible (there is no CPU fallback)")
    dev_index = local_rank % ndev   # one rank per GPU on a real node; rehearsals may share a card
    torch.cuda.set_device(dev_index)
    device = torch.device(f"cuda:{dev_index}")
    # QBOLD_FORCE_PG=1: build the process group for a single rank too, so that a one-GPU box executes the
    # very RCCL path of the N-rank job (communicator on device_id, async all-reduce ring on device buffers)
    use_pg = world > 1 or os.environ.get("QBOLD_FORCE_PG", "0") not in ("", "0")
    if use_pg:
        # RCCL ("nccl" on ROCm) over xGMI; QBOLD_DIST_BACKEND=gloo only for single-card rehearsals
        backend = os.environ.get("QBOLD_DIST_BACKEND", "nccl")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    S, K = args.mc_samples, args.kl_samples
    spec = Spec(args.config, args.protocol, args.voxels, args.encoder_precision, args.tissue, args.exact_f32)
    m = run_workload(spec, args, args.steps, args.warmup, rank, device, use_pg, world, S, K)
    n, T = m["n"], m["T"]
    kernel_ms = m["enc_kernel_ms"] if m["two_launch"] else m["step_kernel_ms"]   # the dominant kernel's launch
    # every rank's own wall time and kernel time; the step time of the job is the MAX over ranks
    mine = torch.tensor([m["elapsed"] / args.steps * 1e3, kernel_ms], dtype=torch.float64, device=device)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if use_pg:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    per_rank = torch.stack(per_rank).cpu().numpy()
    elapsed = float(per_rank[:, 0].max()) * 1e-3 * args.steps
    s = m["sums"]
    neg_elbo = float((s[0] + s[1]) / s[2])
    if not np.isfinite(neg_elbo):
        raise SystemExit("non-finite ELBO")  # the reference's TerminateOnNaN (train.py:375)
    if use_pg and abs(float(s[2]) - float(n) * world) > 0.5:
        raise SystemExit(f"all-reduce of the masked sums is wrong: sum(mask) = {s[2]} for {n} x {world} voxels")

    if rank == 0:
        total_vox = n * world
        value = total_vox * args.steps / elapsed
        default_headline = (args.config == 2 and args.protocol == 11 and args.encoder_precision == "f32"
                            and args.tissue == "table" and not args.exact_f32 and n == 1 << 20)
        line = {
            "metric": "voxel-ELBO evals/sec", "value": value, "unit": "voxel-ELBO evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if args.encoder_precision == "f32"
                      else "bf16 encoder products / f32 accumulate, sampling and ELBO"),
            "data": "synthetic",
            "config": {"workload": workload_text(spec, m, S, K),
                       "global_voxels": total_vox, "parallelism": f"voxel-shard x{world}",
                       "collective": "all_reduce(3 x f64)/step, overlapped with the next step" if use_pg else "none"},
            "neg_elbo": neg_elbo,
            "sum_mask": float(s[2]),   # the (all-reduced) sum of the masks: voxels x ranks for the all-ones mask
            "ranks_seen": dist.get_world_size() if use_pg else 1,
            "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if use_pg else "none",
            **({"allreduce_per_timed_step": m["n_allreduce"] / args.steps} if use_pg else {}),
            "rank_ms_per_step": {"min": float(per_rank[:, 0].min()), "max": float(per_rank[:, 0].max())},
            "rank_kernel_ms": {"min": float(per_rank[:, 1].min()), "max": float(per_rank[:, 1].max())},
            **({"ablation": "QBOLD_DEBUG_SKIP=" + os.environ["QBOLD_DEBUG_SKIP"] + " (NOT a benchmark result)"}
               if os.environ.get("QBOLD_DEBUG_SKIP", "0") not in ("", "0") else {}),
            "roofline": roofline_of(spec, m, S, K, m["step_kernel_ms"], kernel_ms),
        }
        if world == 1 and (args.variants or (default_headline and not args.no_variants)):
            # the other configurations the summaries quote, each timed here like the headline (same ramp, HIP
            # events, barrier-bracketed wall clock) so that every quoted number is in this one line
            vsteps = max(args.variant_steps, 20)
            line["variants"] = {}
            w_headline, params_headline = m["w"], m["params"]
            del m
            for name, kw in VARIANTS:
                torch.cuda.empty_cache()
                vs = Spec(**kw)
                try:
                    vm = run_workload(vs, args, vsteps, min(args.warmup, 5), rank, device, use_pg, world, S, K)
                except Exception as e:   # a variant never takes the headline down
                    line["variants"][name] = {"error": repr(e)}
                    continue
                vk = vm["enc_kernel_ms"] if vm["two_launch"] else vm["step_kernel_ms"]
                vms = vm["elapsed"] / vsteps * 1e3
                vsum = vm["sums"]
                r = roofline_of(vs, vm, S, K, vm["step_kernel_ms"], vk)
                line["variants"][name] = {
                    "workload": workload_text(vs, vm, S, K), "steps": vsteps, "ms_per_step": vms,
                    "value": vm["n"] * 1e3 / vms, "unit": "voxel-ELBO evals/s",
                    "neg_elbo": float((vsum[0] + vsum[1]) / vsum[2]),
                    "roofline": {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                   "kernel_ms", "limiter", "issue")}
                                | {"step_kernel_ms": vm["step_kernel_ms"], "hbm_frac": r["hbm"]["frac"],
                                   "two_pipe_frac": r["two_pipe"]["frac"],
                                   **({"three_pass_frac": r["three_pass"]["frac"]} if "three_pass" in r else {})}}
                del vm
            m = {"w": w_headline, "params": params_headline}
            # one fine-tuning step (forward with saved activations, ELBO backward, encoder backward, AdamW) on a
            # 1 M-voxel batch and on the reference's 38 x 25 x 25 x 8 crop batch: SURVEY rows N2 / N1, timed by
            # scripts/bench_train.py's own loop (150 ms ramp, HIP events)
            torch.cuda.empty_cache()
            try:
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts"))
                import bench_train
                line["training_step"] = bench_train.measure(
                    steps=20, S=1, K=70,   # the reference's training defaults (mc_samples 1, 70 KL draws)
                    config_dir=os.path.join(os.path.dirname(os.path.abspath(__file__)), "config"))
            except Exception as e:
                line["training_step"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(m["params"], m["w"], S, K, args.cpu_budget_s)
        print(json.dumps(line), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

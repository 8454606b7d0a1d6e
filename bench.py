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
    """Times the oracle (reference algorithm: literal 129-node Simpson + Bessel J0 per
    (voxel, sample, tau), encoder, logit-Normal KL) on the host cores, on a bounded sample."""
    from oracle.oracle import Oracle, synth_inputs
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    orc = Oracle("f32", params, threads=cores)

    weights_np = dict(weights_np, gate_offset=-3.0,
                      meta=dict(T=11, U=60, L=2, channelwise_gating=True))

    def run(n):
        x, _ = synth_inputs(n, params, seed=2, oracle=orc)
        t0 = time.perf_counter()
        prior, q, sigma = orc.encoder_fwd(weights_np, x)
        zs = orc.philox_normals(1, 0, 0, n, S)
        zk = orc.philox_normals(1, 1, 0, n, K)
        orc.elbo(x, np.ones(n, np.float32), q, prior, sigma, zs, zk)
        return time.perf_counter() - t0

    n0 = 256 * cores
    t_probe = run(n0)
    n = int(min(max(n0, n0 * budget_s / max(t_probe, 1e-3)), 1 << 20))
    n = max(1024, (n // 1024) * 1024)
    t = run(n)
    return {"value": n / t, "unit": "voxel-ELBO evals/s", "cores": cores, "kind": "port",
            "sample": f"{n} voxels x 11 tau, S={S}, K={K}, C oracle (OpenMP), literal Simpson-129 "
                      f"+ Cephes j0f as the reference computes it; {t:.1f} s"}


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
    ap.add_argument("--rehearse_launch", action="store_true",
                    help="run the N-rank launcher, rendezvous, all-reduce ring and JSON line with no kernel in the step "
                         "(host tensors over gloo; reports no throughput) -- the CPU-container check of --gpus N")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_budget_s", type=float, default=15.0)
    args = ap.parse_args()
    if os.environ.get("QBOLD_DEBUG_SKIP", "0") not in ("", "0"):
        os.environ["QBOLD_ALLOW_ABLATION"] = "1"   # the library ignores QBOLD_DEBUG_SKIP without it
        # the ablation hooks of the kernels (phases switched off for timing experiments)
        # must never reach a reported number
        print("bench.py: QBOLD_DEBUG_SKIP is set -- kernels would skip work; this run is an ablation, not a "
              "benchmark", file=sys.stderr)

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
    from qbold_vi_amd.init import init_encoder_weights
    from qbold_vi_amd.ops import EncoderWeights

    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: no ROCm device is visible (there is no CPU fallback)")
    dev_index = local_rank % ndev   # one rank per GPU on a real node; rehearsals may share a card
    torch.cuda.set_device(dev_index)
    device = torch.device(f"cuda:{dev_index}")
    if world > 1:
        # RCCL ("nccl" on ROCm) over xGMI; QBOLD_DIST_BACKEND=gloo only for single-card rehearsals
        backend = os.environ.get("QBOLD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "config"))
    params = dict(cfg["DEFAULT"])
    T, U, L = 11, 60, 2  # configurations/optimal.yaml
    if args.config == 3:  # 64 taus from -0.015 s in 1.25 ms steps (se_idx 12), width 256
        params.update(tau_start="-0.015", tau_end="0.065", tau_step="0.00125")
        T, U = 64, 256
    elif args.protocol == 24:
        params.update(tau_start="-0.028", tau_end="0.065", tau_step="0.004")
        T = 24
    S, K, n = args.mc_samples, args.kl_samples, args.voxels

    ctx, x = make_inputs(n, params, seed=1 + rank, device=device)
    ctx.set_tissue_mode(args.tissue)
    w = init_encoder_weights(T=T, U=U, L=L, channelwise_gating=True, resid_init_std=0.05,
                             im_loss_sigma=0.05, seed=1)
    if args.encoder_precision == "bf16" and args.config != 2:
        raise SystemExit("--encoder_precision bf16 applies to the fused kernel (config 2)")
    ew = EncoderWeights(ctx, T, U, L, True, -3.0, precision=args.encoder_precision).set_from_arrays(w)
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

    def step(k):
        slot = k % RING
        if pending[slot] is not None:
            pending[slot].wait()   # stream-level wait: the slot's previous all-reduce has finished
            pending[slot] = None
        sums, _, _ = ctx.vi_fwd(ew, x, mask, prior, S, K, seed=1, voxel0=voxel0, out=outs[slot])
        if world > 1:
            pending[slot] = dist.all_reduce(sums, async_op=True)  # sum m*nll, sum kl, sum m
        return sums

    def drain():
        for slot in range(RING):
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp (setup, like input generation above): same work, never timed
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for k in range(8):
            step(k)
        drain()
        torch.cuda.synchronize()
    for k in range(args.warmup):
        step(k)
    drain()
    fence()
    # per-launch duration of the dominant kernel: HIP events on the launch stream
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k, (a, b) in enumerate(evs):
        slot = k % RING
        if pending[slot] is not None:
            pending[slot].wait()
            pending[slot] = None
        a.record()
        sums = ctx.vi_fwd(ew, x, mask, prior, S, K, seed=1, voxel0=voxel0, out=outs[slot])[0]
        b.record()
        if world > 1:
            pending[slot] = dist.all_reduce(sums, async_op=True)
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    # every rank's own wall time and kernel time; the step time of the job is the MAX over ranks
    mine = torch.tensor([elapsed / args.steps * 1e3, kernel_ms], dtype=torch.float64, device=device)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    per_rank = torch.stack(per_rank).cpu().numpy()
    elapsed = float(per_rank[:, 0].max()) * 1e-3 * args.steps
    s = sums.cpu().numpy()
    neg_elbo = float((s[0] + s[1]) / s[2])
    if not np.isfinite(neg_elbo):
        raise SystemExit("non-finite ELBO")  # the reference's TerminateOnNaN (train.py:375)

    if rank == 0:
        total_vox = n * world
        value = total_vox * args.steps / elapsed
        flops = float(algorithmic_flops_per_voxel(T, U, L, S, K)) * n
        byts = algorithmic_bytes_per_voxel(T) * n
        ach_tf = flops / (kernel_ms * 1e-3) / 1e12
        # Two pipes share the launch: the encoder's MACs run on the f16/bf16 matrix pipe (three split-f16
        # passes in f32 mode, one pass in bf16 mode), sampling + ELBO on the f32 vector pipe.  The roof is
        # their serial sum (no overlap assumed), expressed as one composite peak so that frac = achieved/peak.
        enc_flops = 2.0 * encoder_macs_per_voxel(T, U, L) * n
        passes = 1.0 if args.encoder_precision == "bf16" else 3.0
        t_min = (passes * enc_flops / (BF16_MFMA_PEAK_TFLOPS * 1e12) +
                 (flops - enc_flops) / (F32_MFMA_PEAK_TFLOPS * 1e12))
        peak_tf = flops / t_min / 1e12
        frac = ach_tf / peak_tf
        single_pipe = ach_tf / F32_MFMA_PEAK_TFLOPS
        ach_gbs = byts / (kernel_ms * 1e-3) / 1e9
        traffic = None
        counters = {}
        mix = os.path.join(ROOT, "profiles", "r01_vi_fwd_instruction_mix.json")
        if os.path.exists(mix) and args.config == 2 and args.protocol == 11 and args.tissue == "table" and n == 1 << 20:
            # utilisation counters of the committed rocprofv3 --pmc passes of this same command
            pm = json.load(open(mix)).get("pmc", {})
            try:
                cyc = pm["GRBM_GUI_ACTIVE"]["mean"] / 8.0          # cycles per XCD
                counters = {"mfma_util": pm["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0 / cyc,
                            "valu_issue_util": pm["SQ_ACTIVE_INST_VALU"]["mean"] * 4.0 / 1024.0 / cyc,
                            "source": "profiles/r01_vi_fwd_instruction_mix.json"}
            except (KeyError, ZeroDivisionError):
                counters = {}
        pmc = os.path.join(ROOT, "profiles", "r01_vi_fwd_pmc.json")
        if os.path.exists(pmc) and args.config == 2 and args.protocol == 11 and args.tissue == "table" and n == 1 << 20:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        line = {
            "metric": "voxel-ELBO evals/sec", "value": value, "unit": "voxel-ELBO evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.encoder_precision == "f32" else "bf16 encoder products / f32 accumulate, sampling and ELBO",
            "data": "synthetic",
            "config": {"workload": f"{n} synthetic voxels/GPU x {T} tau, S={S} likelihood draws, "
                                   f"K={K} KL draws, optimal.yaml encoder (U={U}, L={L}), fused "
                                   f"qbold_vi_fwd, tissue integral: {args.tissue}, encoder arithmetic: "
                                   f"{args.encoder_precision}",
                       "global_voxels": total_vox, "parallelism": f"voxel-shard x{world}",
                       "collective": "all_reduce(3 x f64)/step, overlapped with the next step" if world > 1 else "none"},
            "neg_elbo": neg_elbo,
            "ranks_seen": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else "none",
            "rank_ms_per_step": {"min": float(per_rank[:, 0].min()), "max": float(per_rank[:, 0].max())},
            "rank_kernel_ms": {"min": float(per_rank[:, 1].min()), "max": float(per_rank[:, 1].max())},
            **({"ablation": "QBOLD_DEBUG_SKIP=" + os.environ["QBOLD_DEBUG_SKIP"] + " (NOT a benchmark result)"}
               if os.environ.get("QBOLD_DEBUG_SKIP", "0") not in ("", "0") else {}),
            "roofline": {"kernel": "vi_fwd_kernel" if args.config == 2 else "wide_dense_kernel (one launch per layer) + elbo_fwd_generic_kernel", "bound": "mfma", "achieved": ach_tf,
                         "peak": peak_tf, "unit": "TFLOP/s",
                         "frac": frac, "traffic": traffic,
                         "kernel_ms": kernel_ms,
                         **({"counters": counters} if counters else {}),
                         "algorithmic_flops_per_voxel": algorithmic_flops_per_voxel(T, U, L, S, K),
                         "peak_components": {"f32_matrix_or_packed_vector_tflops": F32_MFMA_PEAK_TFLOPS,
                                             "f16_bf16_mfma_tflops": BF16_MFMA_PEAK_TFLOPS,
                                             "encoder_mfma_passes": passes,
                                             "encoder_flops_per_voxel": 2 * encoder_macs_per_voxel(T, U, L)},
                         "single_pipe_f32_frac": single_pipe,
                         "note": "compute-bound path on two pipes: 'peak' is the composite of MI355X_MICROARCH.md's "
                                 "dense peaks -- the encoder's flops priced at the f16/bf16 MFMA peak (x the split "
                                 "passes), sampling + ELBO at the f32 matrix (= packed-vector) peak, serial sum, no "
                                 "overlap assumed; 'single_pipe_f32_frac' prices ALL of SURVEY 8(d)'s algorithmic "
                                 "flops at the f32 peak alone and can pass 1 because the encoder's share runs "
                                 "concurrently on the matrix pipe; the kernel is VALU-issue bound (counters); the "
                                 "metric's HBM view is in 'hbm'",
                         "hbm": {"algorithmic_bytes_per_voxel": algorithmic_bytes_per_voxel(T),
                                 "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ach_gbs / HBM_PEAK_GBS}},
        }
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            line["cpu_baseline"] = cpu_baseline(params, w, S, K, args.cpu_budget_s)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""World-size-2 gloo test of the voxel-shard path on CPU.

The kernels cannot run here, so each rank evaluates ITS shard with the oracle (checker role) using
the global-voxel Philox keying, and the product's host logic (qbold_vi_amd.distributed: shard
ranges, all-reduce of the three masked sums, ELBO from sums) combines them.  The result must equal
the single-process evaluation of the whole batch: this is the property that makes the GPU path
independent of the number of ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs():
    import configparser
    sys.path.insert(0, ROOT)
    from oracle.oracle import Oracle, init_weights, synth_inputs
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "config"))
    params = dict(cfg["DEFAULT"])
    o = Oracle("f32", params, threads=2)
    n = 301
    w = init_weights(T=11, U=16, L=1, seed=2)
    w["gate_offset"] = -3.0
    x, _ = synth_inputs(n, params, seed=2, oracle=o)
    prior, q, sigma = o.encoder_fwd(w, x)
    mask = (np.random.default_rng(3).uniform(size=n) > 0.2).astype(np.float32)
    return o, n, x, mask, q, prior, sigma


def _shard_sums(o, x, mask, q, prior, sigma, a, b, S, K, seed):
    zs = o.philox_normals(seed, 0, a, b - a, S)
    zk = o.philox_normals(seed, 1, a, b - a, K)
    return o.elbo(x[a:b], mask[a:b], q[a:b], prior[a:b], sigma[a:b], zs, zk)["sums"]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from qbold_vi_amd import distributed as qd
    r, w, _ = qd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    o, n, x, mask, q, prior, sigma = _inputs()
    a, b = qd.shard_range(n, rank, world)
    sums = torch.as_tensor(_shard_sums(o, x, mask, q, prior, sigma, a, b, 4, 10, 5))
    qd.allreduce_sums(sums)
    nll, kl, elbo = qd.elbo_from_sums(sums)
    g = torch.full((7,), float(rank + 1), dtype=torch.float64)
    qd.allreduce_mean_(g)
    if rank == 0:
        out.put((sums.numpy().copy(), float(nll), float(kl), float(elbo), g.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_voxel_shards_equal_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    sums, nll, kl, elbo, g = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o, n, x, mask, q, prior, sigma = _inputs()
    full = _shard_sums(o, x, mask, q, prior, sigma, 0, n, 4, 10, 5)
    np.testing.assert_allclose(sums, full, rtol=1e-12)
    assert abs(elbo - (full[0] + full[1]) / full[2]) < 1e-12 * abs(elbo)
    assert abs(nll - full[0] / full[2]) < 1e-12 * abs(nll) and np.isfinite(kl)
    np.testing.assert_allclose(g, 1.5)


def test_single_process_helpers_are_noops():
    from qbold_vi_amd import distributed as qd
    s = torch.tensor([2.0, 4.0, 2.0], dtype=torch.float64)
    assert qd.allreduce_sums(s) is s
    nll, kl, elbo = qd.elbo_from_sums(s)
    assert (float(nll), float(kl), float(elbo)) == (1.0, 2.0, 3.0)


def test_forced_single_rank_group_issues_the_collectives():
    """QBOLD_FORCE_PG=1: a one-rank process group (gloo here, RCCL on the GPU box -- tests/test_gpu_rccl.py)
    goes through the same init / all-reduce calls as the N-rank job and leaves the numbers unchanged."""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "from qbold_vi_amd import distributed as qd\n"
        "r, w, _ = qd.init_from_env(backend='gloo')\n"
        "assert (r, w) == (0, 1) and qd.active() and dist.get_world_size() == 1 and qd.backend_name() == 'gloo'\n"
        "s = torch.tensor([2.0, 4.0, 2.0], dtype=torch.float64)\n"
        "qd.allreduce_sums(s); assert s.tolist() == [2.0, 4.0, 2.0]\n"
        "g = torch.arange(5, dtype=torch.float32)\n"
        "h = qd.allreduce_grad_(g, async_op=True); h.wait(); assert g.tolist() == [0, 1, 2, 3, 4]\n"
        "qd.allreduce_mean_(g); assert g.tolist() == [0, 1, 2, 3, 4]\n"
        "assert qd.STATS == {'allreduce_sums': 1, 'allreduce_grad': 2, 'allreduce_other': 0}, qd.STATS\n"
        "dist.destroy_process_group(); print('ok')\n") % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["QBOLD_FORCE_PG"] = "1"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]

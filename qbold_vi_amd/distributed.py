"""Voxel-shard data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

Every voxel's ELBO term is independent, so the path shards into contiguous voxel
ranges with NO data-path collective; the only exchange is the all-reduce of the three masked sums
(sum m*nll, sum [m>0] kl, sum m) per evaluation -- and, in training, of the gradient blob.  Philox
counters are keyed by the GLOBAL voxel index (voxel0 + i), so results do not depend on the number
of ranks.

A process group exists when WORLD_SIZE > 1, or -- with QBOLD_FORCE_PG=1 -- also for a single rank:
the same communicator set-up, the same collectives on the same device buffers, one participant.  That
is how the one-GPU test box executes the RCCL path (tests/test_gpu_rccl.py).
"""
import os
import socket

import torch
import torch.distributed as dist

# collectives issued through this module since import (the RCCL test reads them)
STATS = {"allreduce_sums": 0, "allreduce_grad": 0, "allreduce_other": 0}


def force_pg():
    return os.environ.get("QBOLD_FORCE_PG", "0") not in ("", "0")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_pg()) and not dist.is_initialized():
        if backend is None:
            # RCCL on a real node; QBOLD_DIST_BACKEND=gloo for rehearsals where several ranks share one card
            backend = os.environ.get("QBOLD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if world == 1:   # forced single-rank group: no launcher has set the rendezvous
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        kw = dict(rank=rank, world_size=world)
        if backend == "nccl":
            ndev = torch.cuda.device_count()
            if ndev == 0:
                raise RuntimeError("backend 'nccl' (RCCL) needs a ROCm device; none is visible")
            torch.cuda.set_device(local_rank % ndev)
            dist.init_process_group(backend, device_id=torch.device(f"cuda:{local_rank % ndev}"), **kw)
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def active():
    """True when collectives must be issued (a process group exists, whatever its size)."""
    return dist.is_available() and dist.is_initialized()


def backend_name():
    if not active():
        return "none"
    b = dist.get_backend()
    return b + (" (RCCL)" if b == "nccl" else "")


def shard_range(n_voxels, rank, world):
    """Contiguous [start, stop) of global voxels owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_voxels), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allreduce_(t, kind="allreduce_other", async_op=False):
    """In-place SUM all-reduce of a tensor (no-op without a process group).  With async_op the work
    handle is returned (None without a group): wait() on it before the tensor is read."""
    if not active():
        return None if async_op else t
    STATS[kind] = STATS.get(kind, 0) + 1
    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
    return work if async_op else t


def allreduce_sums(sums):
    """In-place SUM all-reduce of the float64 [3] masked sums (no-op without a process group)."""
    allreduce_(sums, "allreduce_sums")
    return sums


def allreduce_grad_(flat, async_op=False):
    """In-place SUM all-reduce of the flat gradient blob: every shard's gradient already carries the
    GLOBAL 1 / sum(mask), so shard gradients add up."""
    return allreduce_(flat, "allreduce_grad", async_op=async_op)


def allreduce_mean_(flat, world=None):
    """In-place mean all-reduce of a flat gradient blob (pre-training: per-shard mean losses)."""
    if active():
        allreduce_(flat, "allreduce_grad")
        flat.div_(dist.get_world_size() if world is None else world)
    return flat


def elbo_from_sums(sums):
    """(nll, kl, -ELBO) from the reduced sums: masked means as model.py:566,663; ELBO as train.py:351."""
    nll = sums[0] / sums[2]
    kl = sums[1] / sums[2]
    return nll, kl, nll + kl

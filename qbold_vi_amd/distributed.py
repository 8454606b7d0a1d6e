"""Voxel-shard data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

Every voxel's ELBO term is independent, so the path shards into contiguous voxel
ranges with NO data-path collective; the only exchange is the all-reduce of the three masked sums
(sum m*nll, sum [m>0] kl, sum m) per evaluation -- and, in training, of the gradient blob.  Philox
counters are keyed by the GLOBAL voxel index (voxel0 + i), so results do not depend on the number
of ranks.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # RCCL on a real node; QBOLD_DIST_BACKEND=gloo for rehearsals where several ranks share one card
            backend = os.environ.get("QBOLD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend)
    return rank, world, local_rank


def shard_range(n_voxels, rank, world):
    """Contiguous [start, stop) of global voxels owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_voxels), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allreduce_sums(sums):
    """In-place SUM all-reduce of the float64 [3] masked sums (no-op without a process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums


def allreduce_mean_(flat, world=None):
    """In-place mean all-reduce of a flat gradient blob (training)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size() if world is None else world)
    return flat


def elbo_from_sums(sums):
    """(nll, kl, -ELBO) from the reduced sums: masked means as model.py:566,663; ELBO as train.py:351."""
    nll = sums[0] / sums[2]
    kl = sums[1] / sums[2]
    return nll, kl, nll + kl

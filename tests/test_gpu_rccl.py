"""RCCL executes: the process-group path of bench.py and of the training loops with ONE rank and
backend="nccl" (QBOLD_FORCE_PG=1) -- librccl loads, the communicator is created on device_id, the async
all-reduce ring runs on the device float64[3] sums, the gradient blob is all-reduced.  What a real node adds
is more participants in the same calls (8-GPU runs are the driver's to launch).  Each case runs in a child
process: a process group is per-process state and must not leak into the other GPU tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, QBOLD_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "QBOLD_DIST_BACKEND"):
        env.pop(k, None)
    return env


def test_bench_step_over_a_single_rank_rccl_group():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--voxels",
                        "262144", "--ramp_ms", "20", "--no_cpu_baseline", "--no_variants"],
                       capture_output=True, text=True, cwd=ROOT, env=_env(), timeout=900)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-500:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["backend"] == "nccl (RCCL)" and d["ranks_seen"] == 1 and d["n_gpus"] == 1
    assert d["allreduce_per_timed_step"] == 1.0                       # one all-reduce of the sums per step
    assert d["config"]["collective"].startswith("all_reduce(3 x f64)")
    assert d["value"] > 1e8 and d["neg_elbo"] == d["neg_elbo"]
    # the same step without the group gives the same -ELBO: a one-rank SUM is the identity on the sums
    env = _env()
    env.pop("QBOLD_FORCE_PG")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--voxels",
                         "262144", "--ramp_ms", "20", "--no_cpu_baseline", "--no_variants"],
                        capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    d2 = [json.loads(l) for l in r2.stdout.splitlines() if l.startswith("{")][0]
    assert d2["backend"] == "none" and d2["neg_elbo"] == d["neg_elbo"]


TRAIN = r"""
import json, os, sys
sys.path.insert(0, {root!r})
os.chdir({root!r})
import torch, torch.distributed as dist
from qbold_vi_amd import training, distributed as qd
from qbold_vi_amd.utils import load_arguments
args = load_arguments(["train.py", os.path.join({root!r}, "configurations", "optimal.yaml")], entry="train")
args.update(no_units=24, no_intermediate_layers=1, no_pt_epochs=2, no_ft_epochs=2, save_directory={tmp!r},
            synthetic_voxels=20000, mc_samples=2)
model, trainer, hist = training.train_model(args, pt_sample_size=200, max_ft_steps=6)
assert dist.is_initialized()
out = dict(backend=qd.backend_name(), world=dist.get_world_size(), stats=qd.STATS,
           device=str(trainer.context.device), hist=hist)
dist.barrier(); dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def _run_training(tmp, force):
    env = _env()
    if not force:
        env.pop("QBOLD_FORCE_PG")
        code = TRAIN.replace("assert dist.is_initialized()", "assert not dist.is_initialized()") \
                    .replace("dist.barrier(); dist.destroy_process_group()", "") \
                    .replace("world=dist.get_world_size()", "world=1")
    else:
        code = TRAIN
    r = subprocess.run([sys.executable, "-c", code.format(root=ROOT, tmp=str(tmp))], capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])


def test_training_loops_over_a_single_rank_rccl_group(tmp_path):
    d = _run_training(tmp_path / "pg", True)
    assert d["backend"] == "nccl (RCCL)" and d["world"] == 1 and d["device"].startswith("cuda")
    # pre-training: one gradient all-reduce per step; fine-tuning: sums + gradient per step, sums per validation pass
    assert d["stats"]["allreduce_grad"] >= 2 + 6 and d["stats"]["allreduce_sums"] >= 6 + 10
    ft = [h for h in d["hist"] if "val_elbo" in h]
    assert ft and all(h["val_elbo"] == h["val_elbo"] for h in ft)
    # a one-rank SUM / mean all-reduce is the identity: the run without a group produces the same metrics
    e = _run_training(tmp_path / "nopg", False)
    assert e["backend"] == "none" and sum(e["stats"].values()) == 0
    assert [h.get("loss") for h in e["hist"]] == [h.get("loss") for h in d["hist"]]

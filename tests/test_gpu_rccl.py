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


# ---- N > 1: these switch themselves on when the box shows a second device (a driver node); they skip on one-GPU boxes.
def _n_devices():
    import torch
    return torch.cuda.device_count()    # counts devices without initialising the runtime


needs_two_gpus = pytest.mark.skipif(_n_devices() < 2, reason="needs two ROCm devices: RCCL with more than one participant")


def _env_n():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "QBOLD_DIST_BACKEND", "QBOLD_FORCE_PG"):
        env.pop(k, None)
    return env


@needs_two_gpus
def test_bench_two_ranks_over_rccl():
    """`bench.py --gpus 2` as the driver's scaling run uses it, default backend: two participants in every all-reduce."""
    n = 262144
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
                        "--voxels", str(n), "--ramp_ms", "20"],
                       capture_output=True, text=True, cwd=ROOT, env=_env_n(), timeout=900)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-500:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["ranks_seen"] == 2 and d["n_gpus"] == 2 and d["backend"] == "nccl (RCCL)"
    assert d["config"]["global_voxels"] == 2 * n and d["sum_mask"] == 2.0 * n    # the all-reduced sum of the masks
    assert d["allreduce_per_timed_step"] == 1.0
    assert d["rank_ms_per_step"]["max"] == pytest.approx(d["ms_per_step"]) and d["value"] > 2e8


SHARDS = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import configparser
import numpy as np, torch, torch.distributed as dist
from qbold_vi_amd import distributed as qd
from qbold_vi_amd.init import init_encoder_weights
from qbold_vi_amd.ops import Context, EncoderWeights
rank, world, local = qd.init_from_env()
assert world == 2 and qd.backend_name() == os.environ["QBOLD_EXPECT_BACKEND"], qd.backend_name()
dev = torch.device("cuda", torch.cuda.current_device())
cfg = configparser.ConfigParser(); cfg.read(os.path.join({root!r}, "config")); params = dict(cfg["DEFAULT"])
ctx = Context(params, True, True, device=dev)
n, S, K = 100003, 4, 70           # an odd voxel count: the shards differ in size
g = torch.Generator().manual_seed(5)
y = torch.stack([torch.rand(n, generator=g) * 0.6 + 0.1, torch.rand(n, generator=g) * 0.1 + 0.01], -1).to(dev)
x = ctx.signal_fwd(y) * (1.0 + 0.01 * torch.randn(n, 11, generator=g).to(dev))
mask = (torch.rand(n, generator=g) > 0.2).float().to(dev)
w = init_encoder_weights(T=11, U=60, L=2, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=1)
ew = EncoderWeights(ctx, 11, 60, 2, True, -3.0).set_from_arrays(w)
prior, _, _ = ctx.encoder_fwd(ew, x, want=("out1",))
a, b = qd.shard_range(n, rank, world)
sums, q, nk = ctx.vi_fwd(ew, x[a:b].contiguous(), mask[a:b].contiguous(), prior[a:b].contiguous(), S, K, seed=3, voxel0=a)
sums = sums.clone(); qd.allreduce_sums(sums)
sizes = [qd.shard_range(n, r, world)[1] - qd.shard_range(n, r, world)[0] for r in range(world)]
mine = torch.zeros((max(sizes), 2), device=dev)     # all_gather wants equal shapes: pad to the larger shard
mine[:b - a] = nk
parts = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(parts, mine)
parts = [p[:k] for p, k in zip(parts, sizes)]
if rank == 0:
    s1, q1, nk1 = ctx.vi_fwd(ew, x, mask, prior, S, K, seed=3, voxel0=0)     # the one-rank job on the whole batch
    same = bool(torch.equal(torch.cat(parts), nk1))
    rel = float(((sums - s1).abs() / s1.abs()).max())
    print("RESULT " + json.dumps(dict(same=same, rel=rel, sum_mask=float(sums[2]), want_mask=float(mask.sum()),
                                      stats=qd.STATS, world=dist.get_world_size())))
dist.barrier(); dist.destroy_process_group()
"""


def _run_shards(tmp_path, env, port):
    script = tmp_path / "shards.py"
    script.write_text(SHARDS.format(root=ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    # per-voxel outputs bit-equal; the sums differ by the float32 lane partials' order (another tile-to-lane dealing)
    assert d["world"] == 2 and d["same"] and d["rel"] < 1e-8 and d["sum_mask"] == d["want_mask"]
    assert d["stats"]["allreduce_sums"] == 1


@needs_two_gpus
def test_two_rank_shards_equal_the_one_rank_job(tmp_path):
    """Contiguous voxel shards on two GPUs, sums all-reduced over RCCL: per-voxel outputs bit-equal to the one-launch
    job (Philox keyed by the global voxel index), reduced sums equal to summation order."""
    _run_shards(tmp_path, dict(_env_n(), QBOLD_EXPECT_BACKEND="nccl (RCCL)"), 29541)


def test_two_rank_shards_on_one_card_over_gloo(tmp_path):
    """The same script with both ranks on this box's one card and gloo carrying the collectives: what a one-GPU box
    can run of the test above (the shard arithmetic, the gather, the comparison), so that it is known to work the day
    a second device lets the RCCL form run."""
    _run_shards(tmp_path, dict(_env_n(), QBOLD_DIST_BACKEND="gloo", QBOLD_EXPECT_BACKEND="gloo"), 29545)


@needs_two_gpus
def test_two_rank_training_over_rccl_matches_single_process(tmp_path):
    """train.py under the driver's launch line with two GPUs and the default backend (RCCL): voxel shards per rank,
    all-reduced sums and gradients; every rank applies the same update, so the weights equal the single-process
    run's up to the summation order of the gradient all-reduce (the gloo form of this test runs on one card:
    tests/test_gpu_train.py::test_two_rank_training_matches_single_process)."""
    import numpy as np
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configurations", "optimal.yaml")))
    cfg.update(no_pt_epochs=2, no_ft_epochs=2, no_units=16, no_intermediate_layers=1)
    runs, logs = {}, {}
    for name, launcher in (("one", [sys.executable]),
                           ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                    "--master-addr", "127.0.0.1", "--master-port", "29543"])):
        c = dict(cfg, save_directory=str(tmp_path / name))
        ypath = tmp_path / f"{name}.yaml"
        yaml.safe_dump(c, open(ypath, "w"))
        r = subprocess.run(launcher + [os.path.join(ROOT, "train.py"), str(ypath), "--synthetic_voxels", "4096"],
                           cwd=ROOT, env=_env_n(), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        runs[name] = np.load(tmp_path / name / "final_model.npz")
        logs[name] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    for k in runs["one"].files:
        a, b = runs["one"][k], runs["two"][k]
        assert np.isfinite(b).all()
        np.testing.assert_allclose(b, a, rtol=5e-3, atol=5e-4, err_msg=k)
    la = [h["loss"] for h in logs["one"] if "loss" in h]
    lb = [h["loss"] for h in logs["two"] if "loss" in h]
    assert la and len(la) == len(lb)
    np.testing.assert_allclose(lb, la, rtol=2e-3, atol=2e-3)

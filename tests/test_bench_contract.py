"""bench.py's one-line JSON contract: the keys the driver reads, the roofline and cpu_baseline objects, and
the arithmetic that ties them together -- on the committed profile line (CPU) and on a live short run (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
       "vs_baseline", "dtype", "data", "config", "roofline")
ROOF = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms")
CPU = ("value", "unit", "cores", "kind", "sample")


def check_line_legacy(d, expect_cpu_baseline):
    """The roofline object of rounds 1-3 (a composite flop peak at the top level): the committed lines of those rounds."""
    for k in TOP:
        assert k in d, k
    assert d["metric"] == "voxel-ELBO evals/sec" and d["unit"] == "voxel-ELBO evals/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = units all ranks processed / wall time of the timed steps
    assert abs(d["value"] - d["config"]["global_voxels"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ROOF:
        assert k in r, k
    # the regime the counters show: the fused config-2 kernel is bound by the vector pipe's issue rate, the config-3
    # encoder by the f16 matrix pipe
    assert r["bound"] in ("hbm", "mfma", "valu-issue") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
    # achieved = algorithmic flops per launch / the kernel's average launch duration
    per_gpu = d["config"]["global_voxels"] // d["n_gpus"]
    assert abs(r["achieved"] - r["algorithmic_flops_per_voxel"] * per_gpu / (r["kernel_ms"] * 1e-3) / 1e12) \
        < 1e-6 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02   # bracketed steps carry their events' barrier packets
    assert r["traffic"] is None or r["traffic"] > 0
    if "counters" in r:   # only quoted when measured on this build's kernel sources
        assert r["counters"]["source"].startswith("profiles/") and r["counters"]["kernel"]
    st = r["step"]
    assert st["kernel_ms"] >= r["kernel_ms"] * 0.999 and abs(st["frac"] - st["achieved"] / st["peak"]) < 1e-9
    if expect_cpu_baseline:
        c = d["cpu_baseline"]
        for k in CPU:
            assert k in c, k
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
        assert c["unit"] == d["unit"]
        # SURVEY 8(d)'s three cases and the reference-granularity baseline
        assert set(c["cases"]) == {"i_forward_model_only", "ii_elbo_reference_defaults_S1_K70", "iii_elbo_bench_workload"}
        assert all(v["value"] > 0 for v in c["cases"].values())
        assert c["op_granularity"].get("value", 0) > 0, c["op_granularity"]
    if "variants" in d:
        check_variants_legacy(d)


LEGACY_VARIANT_NAMES = {"config3_64tau_width256", "bf16_encoder", "voxels_4194304", "protocol_24tau", "exact_f32_encoder"}


def check_variants_legacy(d):
    """Every configuration a summary quotes is timed in the default run and embedded in the ONE line."""
    v = d["variants"]
    assert set(v) == LEGACY_VARIANT_NAMES
    for name, e in v.items():
        assert "error" not in e, (name, e)
        assert e["steps"] >= 20 and e["ms_per_step"] > 0 and e["unit"] == d["unit"]
        n = 4194304 if name == "voxels_4194304" else 1 << 20
        assert str(n) in e["workload"]
        assert abs(e["value"] - n * 1e3 / e["ms_per_step"]) < 1e-6 * e["value"]
        r = e["roofline"]
        for k in ("kernel", "frac", "kernel_ms", "traffic", "bound", "achieved", "peak"):
            assert k in r, (name, k)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
        assert r["kernel_ms"] <= e["ms_per_step"] * 1.02 and 0.0 < r["hbm_frac"] < 1.0
        assert e["neg_elbo"] == e["neg_elbo"]
    assert v["config3_64tau_width256"]["roofline"]["kernel"].startswith("wide_fused_kernel")
    assert v["config3_64tau_width256"]["roofline"]["bound"] == "mfma"
    assert "exact float32" in v["exact_f32_encoder"]["workload"]
    # the bf16 mode and the exact-f32 mode evaluate the headline's inputs: -ELBO within the re-stated tolerances
    assert abs(v["bf16_encoder"]["neg_elbo"] / d["neg_elbo"] - 1) < 1e-3
    assert abs(v["exact_f32_encoder"]["neg_elbo"] / d["neg_elbo"] - 1) < 1e-5
    # the split-f16 headline beats the strictly-float32 encoder, and 4 M voxels run at the 1 M rate
    assert v["exact_f32_encoder"]["ms_per_step"] > d["ms_per_step"]
    assert 0.8 < v["voxels_4194304"]["value"] / d["value"] < 1.25


def check_cpu_baseline(d):
    c = d["cpu_baseline"]
    for k in CPU:
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert c["unit"] == d["unit"]
    # SURVEY 8(d)'s three cases and the reference-granularity baseline
    assert set(c["cases"]) == {"i_forward_model_only", "ii_elbo_reference_defaults_S1_K70", "iii_elbo_bench_workload"}
    assert all(v["value"] > 0 for v in c["cases"].values())
    assert c["op_granularity"].get("value", 0) > 0, c["op_granularity"]


def check_three_figures(r, n, kernel_ms_step):
    """hbm / issue / two_pipe: each reproducible by its one formula from what the line itself carries."""
    h = r["hbm"]
    assert abs(h["achieved"] - h["algorithmic_bytes_per_voxel"] * n / (kernel_ms_step * 1e-3) / 1e9) < 1e-6 * h["achieved"]
    assert h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12 and 0.0 < h["frac"] < 1.0
    t = r["two_pipe"]
    enc, tot = t["encoder_flops_per_voxel"], t["survey_flops_per_voxel"]
    pk = t["peaks_tflops"]
    assert pk == {"f16_bf16_mfma_dense": 2500.0, "f32_vector": 157.3}
    floor = (enc / 2500e12 + (tot - enc) / 157.3e12) * n
    if "f16_mfma_peak" in t["formula"]:      # no x 3: the split passes are this implementation's cost, not required work
        assert abs(t["frac"] - floor / (kernel_ms_step * 1e-3)) < 1e-9
    assert 0.0 < t["frac"] <= 1.0 and abs(t["survey_tflops"] - tot * n / (kernel_ms_step * 1e-3) / 1e12) < 1e-6 * t["survey_tflops"]
    i = r["issue"]
    if i is not None:     # only with a counter file measured on this build's kernel sources
        c = r["counters"]
        simd = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024
        assert abs(i["frac"] - c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd) < 1e-9 and 0.0 < i["frac"] <= 1.0
        assert i["source"].startswith("profiles/")


def check_line(d, expect_cpu_baseline):
    for k in TOP:
        assert k in d, k
    assert d["metric"] == "voxel-ELBO evals/sec" and d["unit"] == "voxel-ELBO evals/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = units all ranks processed / wall time of the timed steps
    assert abs(d["value"] - d["config"]["global_voxels"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ROOF:
        assert k in r, k
    per_gpu = d["config"]["global_voxels"] // d["n_gpus"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
    if r["bound"] == "hbm":    # the roofline the metric names: algorithmic bytes per launch / launch duration vs 8 TB/s
        assert r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["achieved"] - r["algorithmic_bytes_per_voxel"] * per_gpu / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    else:                      # config 3: the one-launch encoder against the guide's dense f16 MFMA peak (no / 3)
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
        assert abs(r["achieved"] - r["algorithmic_flops_per_voxel"] * per_gpu / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
        assert abs(r["three_pass"]["frac"] - 3.0 * r["frac"]) < 1e-9
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02   # bracketed steps carry their events' barrier packets
    assert r["traffic"] is None or r["traffic"] > 0
    if "counters" in r:   # only quoted when measured on this build's kernel sources
        assert r["counters"]["source"].startswith("profiles/") and r["counters"]["kernel"]
    assert r["step"]["kernel_ms"] >= r["kernel_ms"] * 0.999
    check_three_figures(r, per_gpu, r["step"]["kernel_ms"])
    if expect_cpu_baseline:
        check_cpu_baseline(d)
    if "variants" in d:
        check_variants(d)


VARIANT_NAMES = {"config3_64tau_width256", "bf16_encoder", "voxels_4194304", "protocol_24tau", "exact_f32_encoder",
                 "general_kl_loop_wide_posteriors", "off_grid_spin_echo"}


def check_variants(d):
    """Every configuration a summary quotes is timed in the default run and embedded in the ONE line."""
    v = d["variants"]
    assert set(v) == VARIANT_NAMES
    for name, e in v.items():
        assert "error" not in e, (name, e)
        assert e["steps"] >= 20 and e["ms_per_step"] > 0 and e["unit"] == d["unit"]
        n = 4194304 if name == "voxels_4194304" else 1 << 20
        assert str(n) in e["workload"]
        assert abs(e["value"] - n * 1e3 / e["ms_per_step"]) < 1e-6 * e["value"]
        r = e["roofline"]
        for k in ("kernel", "frac", "kernel_ms", "traffic", "bound", "achieved", "peak", "limiter", "hbm_frac", "two_pipe_frac"):
            assert k in r, (name, k)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
        assert r["kernel_ms"] <= e["ms_per_step"] * 1.02 and 0.0 < r["hbm_frac"] < 1.0 and 0.0 < r["two_pipe_frac"] <= 1.0
        assert e["neg_elbo"] == e["neg_elbo"]
    assert v["config3_64tau_width256"]["roofline"]["kernel"].startswith("wide_fused_kernel")
    assert v["config3_64tau_width256"]["roofline"]["bound"] == "mfma"
    assert "exact float32" in v["exact_f32_encoder"]["workload"]
    # the bf16 mode and the exact-f32 mode evaluate the headline's inputs: -ELBO within the re-stated tolerances
    assert abs(v["bf16_encoder"]["neg_elbo"] / d["neg_elbo"] - 1) < 1e-3
    assert abs(v["exact_f32_encoder"]["neg_elbo"] / d["neg_elbo"] - 1) < 1e-5
    # the split-f16 headline beats the strictly-float32 encoder, and 4 M voxels run at the 1 M rate
    assert v["exact_f32_encoder"]["ms_per_step"] > d["ms_per_step"]
    assert 0.8 < v["voxels_4194304"]["value"] / d["value"] < 1.25
    # the slow sides of the data-dependent switches cost time, and say so in their workload text
    assert v["general_kl_loop_wide_posteriors"]["ms_per_step"] > d["ms_per_step"]
    assert v["off_grid_spin_echo"]["ms_per_step"] > d["ms_per_step"]
    assert "general" in v["general_kl_loop_wide_posteriors"]["workload"] and "off tau = 0" in v["off_grid_spin_echo"]["workload"]


def check_training_step(d):
    """One fine-tuning step on a voxel batch and on the reference's crop batch, timed in the default run too
    (round 3: 2.68 -> ~2.55 ms and 2.62 -> ~1.65 ms; the bounds leave room for the slower boxes of the pool)."""
    t = d["training_step"]
    assert "error" not in t, t
    assert t["voxel_batch"]["voxels"] == 1 << 20 and 0.0 < t["voxel_batch"]["ms_per_step"] < 2.9
    assert t["crop_batch"]["crops"] == [38, 25, 25, 8] and 0.0 < t["crop_batch"]["ms_per_step"] < 1.9


def test_committed_bench_line_keeps_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
    check_line_legacy(d, expect_cpu_baseline=True)
    assert d["n_gpus"] == 1 and d["config"]["global_voxels"] == 1 << 20 and d["dtype"] == "f32"
    assert d["roofline"]["bound"] == "valu-issue" and d["roofline"]["kernel"] == "vi_fwd_kernel"
    # config 3's line: the dominant kernel is the one-launch encoder, timed inside the step
    d3 = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_config3.json")))
    check_line_legacy(d3, expect_cpu_baseline=False)
    assert d3["roofline"]["bound"] == "mfma" and d3["roofline"]["kernel"].startswith("wide_fused_kernel")
    assert d3["roofline"]["kernel_ms"] < d3["roofline"]["step"]["kernel_ms"]


def test_committed_round3_line_carries_the_variants():
    """profiles/r03_bench.json: the default N = 1 run of this round -- headline keys as the contract has them, plus one
    timed entry per configuration the summaries quote."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))
    check_line_legacy(d, expect_cpu_baseline=True)
    assert "variants" in d and d["n_gpus"] == 1 and d["config"]["global_voxels"] == 1 << 20 and d["dtype"] == "f32"
    assert d["roofline"]["kernel"] == "vi_fwd_kernel" and d["roofline"]["bound"] == "valu-issue"
    assert d["value"] > 2.0e9                                   # round 2's driver-witnessed headline: 2.04e9
    check_training_step(d)
    assert 0.015 < d["roofline"]["hbm"]["frac"] < 0.05          # the metric's HBM roofline: ~2 - 3 % by arithmetic
    d3 = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_config3.json")))
    check_line_legacy(d3, expect_cpu_baseline=False)
    assert d3["roofline"]["bound"] == "mfma" and d3["ms_per_step"] < 4.0


def test_bench_refuses_to_run_without_a_gpu():
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_self_launcher_starts_the_ranks_and_prints_one_line():
    """`python3 bench.py --gpus 2` with no torchrun around it: the parent spawns the ranks before anything
    touches a GPU, the ranks rendezvous (gloo here), run the all-reduce ring and rank 0 prints ONE line that
    says how many ranks the process group saw.  --rehearse_launch puts no kernel in the step (no GPU here)."""
    env = dict(os.environ, QBOLD_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--rehearse_launch"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["steps"] == 6
    assert d["value"] is None and "rehearsal" in d          # no kernel ran: no throughput is claimed
    assert d["rank_ms_per_step"]["min"] <= d["rank_ms_per_step"]["max"] == d["ms_per_step"]


def test_driver_launch_line_reaches_the_same_path():
    """The driver's own N > 1 command (torch.distributed.run around bench.py) must not spawn a second layer."""
    env = dict(os.environ, QBOLD_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse_launch"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2


def test_self_launcher_fails_when_a_rank_fails():
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode != 0 and not _json_lines(r.stdout)


@pytest.mark.gpu
def test_self_launched_two_ranks_on_one_card():
    """The real step under the self-launcher: two ranks share this box's one card (gloo carries the three sums;
    on a node each rank has its own GPU and the backend is RCCL)."""
    env = dict(os.environ, QBOLD_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--voxels", "65536", "--ramp_ms", "20"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    check_line(d, expect_cpu_baseline=False)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["config"]["global_voxels"] == 2 * 65536
    assert d["rank_ms_per_step"]["max"] == pytest.approx(d["ms_per_step"])


@pytest.mark.gpu
def test_driver_command_line_carries_every_variant():
    """The driver's own N = 1 command: headline keys unchanged, and the variants object beside them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--cpu_budget_s", "1"], capture_output=True, text=True, cwd=ROOT, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_line(d, expect_cpu_baseline=True)
    assert "variants" in d and d["config"]["global_voxels"] == 1 << 20 and d["dtype"] == "f32"
    check_training_step(d)


@pytest.mark.gpu
def test_live_bench_line_keeps_the_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--voxels",
                        "65536", "--ramp_ms", "20", "--cpu_budget_s", "1"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # ONE JSON line
    d = json.loads(lines[0])
    check_line(d, expect_cpu_baseline=True)
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1

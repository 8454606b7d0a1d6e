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
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
    # achieved = algorithmic flops per launch / the kernel's average launch duration
    per_gpu = d["config"]["global_voxels"] // d["n_gpus"]
    assert abs(r["achieved"] - r["algorithmic_flops_per_voxel"] * per_gpu / (r["kernel_ms"] * 1e-3) / 1e12) \
        < 1e-6 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert r["traffic"] is None or r["traffic"] > 0
    if expect_cpu_baseline:
        c = d["cpu_baseline"]
        for k in CPU:
            assert k in c, k
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
        assert c["unit"] == d["unit"]


def test_committed_bench_line_keeps_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))
    check_line(d, expect_cpu_baseline=True)
    assert d["n_gpus"] == 1 and d["config"]["global_voxels"] == 1 << 20 and d["dtype"] == "f32"
    # the PMC traffic of the profiled launch sits at the algorithmic bytes (nothing re-read, no spill traffic)
    alg = d["roofline"]["hbm"]["algorithmic_bytes_per_voxel"] * (1 << 20)
    assert alg <= d["roofline"]["traffic"] < 1.25 * alg


def test_bench_refuses_to_run_without_a_gpu():
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


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

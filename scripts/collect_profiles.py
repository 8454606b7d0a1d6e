#!/usr/bin/env python3
"""Copy the summaries of the round's profiling calls (scripts/profile_round2.sh a / b, all under
gpurun_out/profiles_r02/) into profiles/ -- the tracked copies the numbers in DESIGN.md and README.md come
from -- and derive the per-kernel counter files bench.py quotes (roofline.traffic / roofline.counters).  Every
derived file carries the fingerprint of the kernel sources it was measured on (qbold_vi_amd.build.
source_fingerprint, recorded on the GPU box by the profiling script); bench.py drops a file whose fingerprint
is not the current build's."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(G, f"profiles_{tag}")
NOTE = ("traffic = 2*FETCH_SIZE + WRITE_SIZE KiB (gfx950 correction of MI355X_MICROARCH.md, HBM section; the x2 is "
        "calibrated there for 16 B/lane streams, so for dword-row reads the read side is an upper bound); FETCH_SIZE "
        "and WRITE_SIZE from separate rocprofv3 --pmc passes of bench.py, per-dispatch means over the timed launches")


def last_json_line(path):
    with open(path) as f:
        rows = [l for l in f if l.startswith("{") and '"metric"' in l]
    return json.loads(rows[-1]) if rows else None


def sha(part):
    """What the profiling script recorded on the GPU box: the digests of every source file (round 4 on:
    qbold_vi_amd.build.file_digests), or one fingerprint over all sources (rounds 2 and 3)."""
    p = os.path.join(src, f"source_sha256_{part}.txt")
    if not os.path.exists(p):
        return None
    text = open(p).read().strip()
    return json.loads(text) if text.startswith("{") else text


def unit_sha(recorded, units):
    """(fingerprint, units) over the translation units a kernel lives in and every header: valid while other units change."""
    if recorded is None or isinstance(recorded, str):
        return recorded, None
    sys.path.insert(0, ROOT)
    from qbold_vi_amd.build import source_fingerprint
    return source_fingerprint(units, recorded), units


def pmc_file(summary_name, kernel, n, algorithmic, recorded, units, extra=None):
    fingerprint, units = unit_sha(recorded, units)
    s = json.load(open(os.path.join(src, summary_name)))
    c = {k: v["mean"] for k, v in s["pmc"].items()}
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    d = {"kernel": kernel, "voxels": n, "source_sha256": fingerprint, "source_units": units,
         "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
         "hbm_read_bytes_x2_gfx950": 2 * c["FETCH_SIZE"] * 1024, "hbm_write_bytes": c["WRITE_SIZE"] * 1024,
         "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
         "algorithmic_bytes_per_launch": algorithmic,
         "counters": {k: v for k, v in c.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")},
         "dispatch": s.get("dispatch"), "kernel_stats": s.get("kernel_stats"), "note": NOTE}
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c:
        pass
    d.update(extra or {})
    return d


def put(name, obj):
    if obj is not None:
        json.dump(obj, open(os.path.join(P, name), "w"), indent=1)


n = 1 << 20
copied = []
for name in sorted(os.listdir(src)):
    if name.startswith("source_sha256") or (name.startswith(f"{tag}_train_") and name.endswith("_summary.json")):
        continue
    dst = os.path.join(P, name)
    if name.startswith(f"{tag}_bench") and name.endswith(".json"):
        line = last_json_line(os.path.join(src, name))
        if line:
            json.dump(line, open(dst, "w"))
            copied.append(name)
    else:
        shutil.copy(os.path.join(src, name), dst)
        copied.append(name)
fa, fb = sha("a"), sha("b")
if fa and os.path.exists(os.path.join(src, f"{tag}_vi_fwd_summary.json")):
    kname = "vi_fwd_kernel<11,2,2,true,false,false" + (",true,false>" if tag >= "r04" else ",true>" if tag >= "r03" else ">")
    put(f"{tag}_vi_fwd_pmc.json", pmc_file(f"{tag}_vi_fwd_summary.json", kname, n, 96 * n, fa, ["vi_kernels.hip"]))
if fb:
    if os.path.exists(os.path.join(src, f"{tag}_config3_wide_fused_summary.json")):
        T = 64
        enc = pmc_file(f"{tag}_config3_wide_fused_summary.json", "wide_fused_kernel<4, 2, true>", n, (4 * T + 4 * T + 20) * n, fb, ["elbo_kernels.hip", "wide_fused_kernels.hip"])
        elbo = pmc_file(f"{tag}_config3_elbo_summary.json", ("elbo_fwd_gt64_kernel" if tag >= "r04" else "elbo_fwd_lds_kernel") + "<64, 12, true>", n,
                        (4 * T + 4 * T + 20 + 4 + 20 + 8) * n, fb, ["elbo_kernels.hip"])
        if enc and elbo:
            enc["step"] = {"launches": ["wide_fused_kernel", "elbo_fwd_gt64_kernel" if tag >= "r04" else "elbo_fwd_lds_kernel"],
                           "hbm_bytes": enc["hbm_bytes_per_launch"] + elbo["hbm_bytes_per_launch"],
                           "algorithmic_bytes": (4 * T + 4 + 20 + 20 + 8) * n,
                           "elbo_kernel": {k: elbo[k] for k in ("kernel", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes_per_launch",
                                                                "algorithmic_bytes_per_launch", "counters", "dispatch")},
                           "note": "the step's algorithmic bytes (SURVEY 8d) count signals, mask, prior in and q, (nll, kl) "
                                   "out; the two-launch path also writes and re-reads log sigma (2 x 4T B) and q, and reads "
                                   "the signals twice: its own floor is the sum of the two kernels' algorithmic bytes"}
        put(f"{tag}_config3_pmc.json", enc)
    for t, kernel, alg in (("p24", "vi_fwd_kernel<24,2,7,true,false,false" + (",true,false>" if tag >= "r04" else ">"), (4 * 24 + 52) * n),
                           ("bf16", "vi_fwd_kernel<11,2,2,true,false,true" + (",true,false>" if tag >= "r04" else ",true>" if tag >= "r03" else ">"), 96 * n)):
        if os.path.exists(os.path.join(src, f"{tag}_{t}_vi_fwd_summary.json")):
            put(f"{tag}_{t}_vi_fwd_pmc.json", pmc_file(f"{tag}_{t}_vi_fwd_summary.json", kernel, n, alg, fb, ["vi_kernels.hip"]))
fc = sha("c")
if fc:
    train = {"workload": "scripts/bench_train.py --only voxel: one fine-tuning step on 1,048,576 voxels ([N][64] float32 "
                         "tensors = 268 MB each)", "source_sha256": unit_sha(fc, None)[0], "note": NOTE, "kernels": {}}
    for k in ("block_bwd_dw_kernel", "block_bwd_kernel", "encoder_train_fwd_kernel", "xtd_kernel", "elbo_bwd_kernel"):
        f = os.path.join(src, f"{tag}_train_{k}_summary.json")
        if os.path.exists(f):
            s_ = json.load(open(f))
            c_ = {n: v["mean"] for n, v in s_["pmc"].items()}
            if "FETCH_SIZE" in c_ and "WRITE_SIZE" in c_:
                train["kernels"][k] = {"FETCH_SIZE_KB": c_["FETCH_SIZE"], "WRITE_SIZE_KB": c_["WRITE_SIZE"],
                                       "hbm_read_MB_x2_gfx950": 2 * c_["FETCH_SIZE"] * 1024 / 1e6,
                                       "hbm_write_MB": c_["WRITE_SIZE"] * 1024 / 1e6,
                                       "kernel_stats": [r for r in s_["kernel_stats"] if k in r["name"]]}
    put(f"{tag}_train_pmc.json", train)
print("profiles/ updated:", copied)

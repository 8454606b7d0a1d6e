#!/usr/bin/env python3
"""Copy the summaries of one profiling call (scripts/profile_bench.sh + scripts/pmc_mix.sh + bench.py runs,
all under gpurun_out/) into profiles/ -- the tracked copies the numbers in DESIGN.md come from."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
def last_json_line(path):
    with open(path) as f:
        rows = [l for l in f if l.startswith("{") and '"metric"' in l]
    return json.loads(rows[-1])
src = os.path.join(G, f"profiles_{tag}")
for name in (f"{tag}_kernel_stats.csv", f"{tag}_vi_fwd_summary.json", f"{tag}_bench_under_rocprof.json"):
    shutil.copy(os.path.join(src, name), os.path.join(P, name))
summ = json.load(open(os.path.join(src, f"{tag}_vi_fwd_summary.json")))
fetch_kb = summ["pmc"]["FETCH_SIZE"]["mean"]; write_kb = summ["pmc"]["WRITE_SIZE"]["mean"]
n = 1 << 20
pmc = {"kernel": "vi_fwd_kernel<11,2,2,true,false,false>", "voxels": n,
       "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
       "hbm_read_bytes_raw": fetch_kb * 1024, "hbm_read_bytes_x2_gfx950": 2 * fetch_kb * 1024,
       "hbm_write_bytes": write_kb * 1024, "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
       "algorithmic_bytes_per_launch": 96 * n,
       "note": "traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md, HBM section; the x2 is "
               "calibrated there for 16 B/lane streams, this kernel reads 44-byte rows with dword loads, so the read "
               "side is an upper bound); separate --pmc passes"}
json.dump(pmc, open(os.path.join(P, f"{tag}_vi_fwd_pmc.json"), "w"), indent=1)
shutil.copy(os.path.join(G, "mix.json"), os.path.join(P, f"{tag}_vi_fwd_instruction_mix.json"))
for src_name, dst in (("bench_final.json", f"{tag}_bench.json"), ("bench_bf16.json", f"{tag}_bench_bf16.json"),
                      ("bench_4m.json", f"{tag}_bench_4m.json")):
    p = os.path.join(G, src_name)
    if os.path.exists(p):
        json.dump(last_json_line(p), open(os.path.join(P, dst), "w"))
if os.path.exists(os.path.join(G, "api.json")):
    shutil.copy(os.path.join(G, "api.json"), os.path.join(P, f"{tag}_api_kernels.json"))
print("profiles/ updated:", sorted(os.listdir(P)))

#!/usr/bin/env python3
"""Summarise a scripts/profile_bench.sh output directory: per-kernel stats + mean PMC values."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
kfilter = sys.argv[2] if len(sys.argv) > 2 else "vi_fwd"
out = {"dir": d, "kernel_filter": kfilter, "kernel_stats": [], "pmc": {}}
for f in glob.glob(f"{d}/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("qb::", "anonymous namespace)::")) and "at::" not in r["Name"]:
            nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            out["kernel_stats"].append({"name": nm.split("(")[0][-60:] if "<" not in nm else nm[:90],
                                        "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                        "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])})
for f in glob.glob(f"{d}/pmc_*/*/*_counter_collection.csv"):
    acc = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(f)):
        if kfilter in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                       "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
    for k, v in acc.items():
        out["pmc"][k] = {"n": len(v), "mean": sum(v) / len(v)}
    if meta:
        out["dispatch"] = meta
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Print the bench lines and the per-kernel averages of one gpurun call (files under gpurun_out/).
Usage: scripts/show_run.py [bench json/log ...] [--prof DIR ...]"""
import csv
import glob
import json
import sys


def main():
    args = sys.argv[1:]
    profs, files = [], []
    while args:
        a = args.pop(0)
        if a == "--prof":
            profs.append(args.pop(0))
        else:
            files.append(a)
    for f in files:
        try:
            rows = [l for l in open(f) if l.startswith("{") and '"metric"' in l]
        except OSError as e:
            print(f, e)
            continue
        for l in rows[-1:]:
            d = json.loads(l)
            r = d.get("roofline", {})
            print(f"{f}: value={d['value']:.4g} ms/step={d['ms_per_step']:.4f} kernel_ms={r.get('kernel_ms')} "
                  f"frac={r.get('frac')} bound={r.get('bound')} n_gpus={d['n_gpus']}")
    for p in profs:
        for f in glob.glob(f"{p}/**/*kernel_stats.csv", recursive=True):
            print(f)
            for r in list(csv.DictReader(open(f)))[:16]:
                print(f"  {float(r['AverageNs']) / 1e3:10.1f} us x{r['Calls']:>5}  {r['Name'][:110]}")


if __name__ == "__main__":
    main()

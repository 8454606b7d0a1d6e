#!/usr/bin/env python3
"""Cut one kernel out of a device assembly file (hipcc -S --cuda-device-only) and summarise it:
   scripts/dev/kernel_isa.py file.s <mangled-name-substring> [out.s]
prints instruction-class counts, the scratch accesses with their line offsets, and s_waitcnt counts."""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.endswith(":") is False and re.match(r"^[_A-Za-z0-9]+:\s*(;.*)?$", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(body))
cls = collections.Counter()
for i, l in enumerate(body):
    t = l.strip().split(" ")[0]
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    if t.startswith("scratch_"):
        print(f"{i:6d}  {l.strip()[:100]}")
    k = ("mfma" if "mfma" in t else "trans" if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", t) else
         "valu" if t.startswith("v_") else "salu" if t.startswith("s_") else "lds" if t.startswith("ds_") else
         "vmem" if t.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
    cls[k] += 1
print(dict(cls), "lines", len(body))

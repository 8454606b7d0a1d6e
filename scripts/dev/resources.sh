#!/bin/bash
# Register / scratch / occupancy of the kernels of one translation unit, from the compiler's own report (no GPU):
#   scripts/dev/resources.sh vi_kernels.hip [name-substring] [extra hipcc flags]
cd "$(dirname "$0")/../.."
SRC=$1; PAT=${2:-.}; shift; shift
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-function "$@" \
    -Rpass-analysis=kernel-resource-usage -c qbold_vi_amd/csrc/$SRC -o /tmp/res_$$.o 2>&1 | \
python3 -c '
import re, sys
pat = sys.argv[1]
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "SGPRs", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
        if m and cur: rows[cur][key.split(" ")[0]] = int(m.group(1))
import subprocess
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
    if re.search(pat, name): print(v, name)
' "$PAT"
rm -f /tmp/res_$$.o

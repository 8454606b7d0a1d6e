#!/bin/bash
# Run on the GPU box (via gpurun): HBM traffic (PMC, separate passes) of the training step's GEMM kernels.
# Usage: scripts/profile_train.sh   -> gpurun_out/train_pmc.json
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_train_pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/bench_train.py --only voxel --steps 3 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/bench_train.py --only voxel --steps 3 > $OUT/write.log 2>&1
python3 - "$OUT" <<'PY' > gpurun_out/train_pmc.json
import collections, csv, glob, json, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if any(k in n for k in ("xw64", "xtd_kernel", "xw_kernel", "gate_", "elbo_bwd", "head_delta", "normalise64")):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for n, c in sorted(acc.items()):
    f = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
    w = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
    out[n] = {"launches": len(c["FETCH_SIZE"]), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
              "hbm_read_MB_x2_gfx950": 2 * f * 1024 / 1e6, "hbm_write_MB": w * 1024 / 1e6}
json.dump({"workload": "scripts/bench_train.py --only voxel (1,048,576 voxels, [N][64] float32 tensors = 268 MB each)",
           "note": "per-launch means; read side with the gfx950 x2 correction of MI355X_MICROARCH.md (an upper bound for "
                   "16-byte loads), separate --pmc passes", "kernels": out}, sys.stdout, indent=1)
PY

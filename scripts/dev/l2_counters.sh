#!/bin/bash
# L2 (TCC) hit / miss / request counters of the crop step's kernels: one --pmc pass of scripts/bench_train.py --only crop
export TMPDIR=/tmp
R=$PWD
D=$R/gpurun_out/l2_counters
rm -rf $D; mkdir -p $D
(cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $D/pmc -- python3 $R/scripts/bench_train.py --only crop --steps 3 > $D/run.log 2>&1) || { echo FAILED; tail -5 $D/run.log; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$D/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if any(s in k for s in ("conv9h", "xtd9b", "gate_bwd", "xw64_kernel")):
        print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
rm -rf $D/pmc

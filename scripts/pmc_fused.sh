#!/bin/bash
# PMC passes for the config-3 bench (run on the GPU box via gpurun): scripts/pmc_fused.sh <tag> [bench args]
TAG=${1:-c3}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--config 3 --steps 6 --warmup 2 --ramp_ms 20 --no_cpu_baseline $@"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_a -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/a.log 2>&1 || echo a failed
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/b.log 2>&1 || echo b failed
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/f.log 2>&1 || echo fetch failed
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/w.log 2>&1 || echo write failed
cd $GRAFT_REPO_ROOT
for k in wide_fused elbo_fwd_lds; do python3 scripts/summarise_prof.py $OUT $k > gpurun_out/pmc_${TAG}_$k.json; done
python3 - <<P
import json
for k in ("wide_fused", "elbo_fwd_lds"):
    d = json.load(open("gpurun_out/pmc_${TAG}_%s.json" % k))
    print(k, {n: round(v["mean"]) for n, v in d["pmc"].items()}, d.get("dispatch"))
P

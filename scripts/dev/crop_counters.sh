#!/bin/bash
# SQ / LDS counters of the crop training step's kernels (conv9h_kernel, xtd9b_kernel, ...): two --pmc passes of
# scripts/bench_train.py --only crop, summarised per kernel into gpurun_out/crop_counters/<kernel>.json
export TMPDIR=/tmp
R=$PWD
D=$R/gpurun_out/crop_counters
rm -rf $D; mkdir -p $D
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
B="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/scripts/bench_train.py --only ${ONLY:-crop} --steps 10 > $D/trace.log 2>&1) || echo "FAILED trace"
(cd /tmp && rocprofv3 --pmc $A --output-format csv -d $D/pmc_a -- python3 $R/scripts/bench_train.py --only ${ONLY:-crop} --steps 3 > $D/a.log 2>&1) || echo "FAILED a"
(cd /tmp && rocprofv3 --pmc $B --output-format csv -d $D/pmc_b -- python3 $R/scripts/bench_train.py --only ${ONLY:-crop} --steps 3 > $D/b.log 2>&1) || echo "FAILED b"
for k in "$@"; do python3 scripts/summarise_prof.py $D $k > $D/$k.json; done
rm -rf $D/pmc_a $D/pmc_b $D/trace/*/*_kernel_trace.csv
ls $D

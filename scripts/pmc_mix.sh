#!/bin/bash
# Instruction-mix PMC passes for the bench workload (run on the GPU box via gpurun).
set -e
TAG=${1:-mix}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no_cpu_baseline $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 --output-format csv -d $OUT/pmc_a -- python3 bench.py $ARGS > $OUT/a.log 2>&1 || echo a failed
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_b -- python3 bench.py $ARGS > $OUT/b.log 2>&1 || echo b failed
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_c -- python3 bench.py $ARGS > $OUT/c.log 2>&1 || echo c failed
python3 scripts/summarise_prof.py $OUT vi_fwd

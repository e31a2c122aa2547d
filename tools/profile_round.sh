#!/bin/bash
# every profile a round commits, in one go (through gpurun, from the repo root; ~12 minutes):
#   tools/profile_round.sh r4  ->  gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
set -e
R=${1:-r4}
OUT=gpurun_out
tools/profile_bench.sh f32 > $OUT/${R}_prof_f32.log 2>&1
cp $OUT/kernel_stats_f32.csv $OUT/${R}_fused_f32_kernel_stats.csv; cp $OUT/pmc_summary_f32.json $OUT/${R}_final_pmc_summary.json; cp $OUT/prof_f32_kt_bench.json $OUT/${R}_fused_f32_bench_under_rocprof.json
echo "f32 done"
tools/profile_bench.sh f16 > $OUT/${R}_prof_f16.log 2>&1
cp $OUT/kernel_stats_f16.csv $OUT/${R}_fused_f16_kernel_stats.csv; cp $OUT/pmc_summary_f16.json $OUT/${R}_f16_head_pmc_summary.json; cp $OUT/prof_f16_kt_bench.json $OUT/${R}_fused_f16_bench_under_rocprof.json
echo "f16 done"
TAG=${R}_train_step tools/profile_train.sh > $OUT/${R}_prof_train.log 2>&1
TAG=${R}_train_step_f16rec tools/profile_train.sh --train-records f16 >> $OUT/${R}_prof_train.log 2>&1
TAG=${R}_train_step_f16fwd tools/profile_train.sh --train-records f16 --train-forward f16 >> $OUT/${R}_prof_train.log 2>&1
TAG=${R}_train_step_f16 tools/profile_train.sh --train-records f16 --train-forward f16 --train-backward f16 >> $OUT/${R}_prof_train.log 2>&1
echo "train done"
ls $OUT | grep "^${R}_" | head -40

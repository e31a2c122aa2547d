#!/bin/bash
# rocprofv3 passes over the fused cfg2 frame (tools/cfg2_bench.py, schedule 8x8): kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in their own runs
#   tools/profile_cfg2.sh [f32|f16]     (through gpurun, from the repo root) -> gpurun_out/cfg2_<prec>_*
set -e
export TMPDIR=/tmp
REPO=$(pwd)
PREC=${1:-f32}
OUT=$REPO/gpurun_out
P=$OUT/cfg2_${PREC}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_kt -- python3 $REPO/tools/cfg2_bench.py $PREC 8,8 > ${P}_kt_bench.json 2> ${P}_kt.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${P}_pmc_fetch -- python3 $REPO/tools/cfg2_bench.py $PREC 8,8 > ${P}_pmc_fetch_bench.json 2> ${P}_pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${P}_pmc_write -- python3 $REPO/tools/cfg2_bench.py $PREC 8,8 > ${P}_pmc_write_bench.json 2> ${P}_pmc_write.err
echo "pmc write done"
cd $REPO
python3 tools/summarize_pmc.py ${P}_pmc_fetch ${P}_pmc_write > $OUT/cfg2_pmc_summary_${PREC}.json
find ${P}_kt -name "*kernel_stats.csv" -exec cp {} $OUT/cfg2_kernel_stats_${PREC}.csv \;
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/cfg2_kernel_stats_${PREC}.csv | head -12

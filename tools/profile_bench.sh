#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats        2-4. PMC passes (separate runs, as MI355X_MICROARCH.md prescribes)
# Output lands in gpurun_out/prof_*; copy the summaries you want judged into profiles/.
set -e
export TMPDIR=/tmp
REPO=$(pwd)
ARGS="--steps 5 --warmup 2 --no-clock-probe --no-cpu-baseline --no-grid-roofline --no-fat-schedule --no-fp16-leg --no-occupancy --no-dense192 --no-train"
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -- python3 $REPO/bench.py $ARGS > $OUT/prof_kt_bench.json 2> $OUT/prof_kt.err
echo "kernel trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/prof_pmc_sq_bench.json 2> $OUT/prof_pmc_sq.err
echo "pmc sq done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/prof_pmc_fetch_bench.json 2> $OUT/prof_pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/prof_pmc_write_bench.json 2> $OUT/prof_pmc_write.err
echo "pmc write done"
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_pmc_sq $OUT/prof_pmc_fetch $OUT/prof_pmc_write > $OUT/pmc_summary.json
find $OUT/prof_kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls $OUT

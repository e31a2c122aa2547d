#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile_bench.sh [f32|f16] [extra bench.py args]
#   1. kernel trace + stats        2-5. PMC passes (separate runs, as MI355X_MICROARCH.md prescribes; no trace domains next to --pmc)
# Output lands in gpurun_out/prof_<prec>_*; copy the summaries you want judged into profiles/.
set -e
export TMPDIR=/tmp
REPO=$(pwd)
PREC=${1:-f32}
shift || true
ARGS="--steps 5 --warmup 2 --no-side-legs --precision $PREC $*"
OUT=$REPO/gpurun_out
P=$OUT/prof_${PREC}
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_kt -- python3 $REPO/bench.py $ARGS > ${P}_kt_bench.json 2> ${P}_kt.err
echo "kernel trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d ${P}_pmc_sq -- python3 $REPO/bench.py $ARGS > ${P}_pmc_sq_bench.json 2> ${P}_pmc_sq.err
echo "pmc sq done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d ${P}_pmc_inst -- python3 $REPO/bench.py $ARGS > ${P}_pmc_inst_bench.json 2> ${P}_pmc_inst.err
echo "pmc inst done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${P}_pmc_fetch -- python3 $REPO/bench.py $ARGS > ${P}_pmc_fetch_bench.json 2> ${P}_pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${P}_pmc_write -- python3 $REPO/bench.py $ARGS > ${P}_pmc_write_bench.json 2> ${P}_pmc_write.err
echo "pmc write done"
cd $REPO
python3 tools/summarize_pmc.py ${P}_pmc_sq ${P}_pmc_inst ${P}_pmc_fetch ${P}_pmc_write > $OUT/pmc_summary_${PREC}.json
# per-kernel median / mean without the warm-up launches (rocprofv3's own --stats table averages every launch, warm-ups included)
python3 tools/kernel_medians.py ${P}_kt --skip 2 > $OUT/kernel_stats_${PREC}.csv
cp ${P}_kt_bench.json $OUT/bench_under_rocprof_${PREC}.json
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls $OUT | head -50

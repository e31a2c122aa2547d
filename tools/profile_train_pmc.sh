#!/bin/bash
# rocprofv3 PMC passes over the cfg3 training step (bench.py --train-only): matrix-pipe occupancy, instruction mix and HBM bytes of the
# record-mode kernels.  Extra arguments go to bench.py (e.g. --train-forward f16 --train-backward f16); TAG names the output.
# Separate runs per counter group, no trace domains next to --pmc (MI355X_MICROARCH.md).
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
TAG=${TAG:-train}
ARGS="--train-only --steps 4 --warmup 2 $*"
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_tpmc_${TAG}_sq -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/prof_tpmc_${TAG}.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/prof_tpmc_${TAG}_inst -- python3 $REPO/bench.py $ARGS > /dev/null 2>> $OUT/prof_tpmc_${TAG}.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_tpmc_${TAG}_fetch -- python3 $REPO/bench.py $ARGS > /dev/null 2>> $OUT/prof_tpmc_${TAG}.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_tpmc_${TAG}_write -- python3 $REPO/bench.py $ARGS > /dev/null 2>> $OUT/prof_tpmc_${TAG}.err
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_tpmc_${TAG}_sq $OUT/prof_tpmc_${TAG}_inst $OUT/prof_tpmc_${TAG}_fetch $OUT/prof_tpmc_${TAG}_write > $OUT/${TAG}_pmc_summary.json
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete
echo done

#!/bin/bash
# instruction counters of the frame kernel on rank 0's tile of an N-way sharded frame, per forced samples-per-pass S
#   tools/profile_tile_pmc.sh [f32|f16] [ways] -> gpurun_out/tile_pmc_<prec>_S<S>.json
set -e
export TMPDIR=/tmp
REPO=$(pwd)
PREC=${1:-f16}; WAYS=${2:-8}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
for S in 1 2 4; do
  ARGS="--steps 5 --warmup 2 --no-side-legs --no-clock-probe --precision $PREC --shard-of $WAYS --tiles interleaved --steps-per-pass $S"
  P=$OUT/tilepmc_${PREC}_S$S
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d ${P}_inst -- python3 $REPO/bench.py $ARGS > ${P}_bench.json 2> ${P}.err
  rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d ${P}_sq -- python3 $REPO/bench.py $ARGS > /dev/null 2>> ${P}.err
  python3 $REPO/tools/summarize_pmc.py ${P}_inst ${P}_sq > $OUT/tile_pmc_${PREC}_S$S.json
  echo "S=$S done"
done
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete

#!/bin/bash
# where do the record kernels' waves wait?  issue / memory-FIFO stall counters over bench.py --train-only (extra args go to bench.py)
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
TAG=${TAG:-stalls}
ARGS="--train-only --steps 4 --warmup 2 $*"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/prof_${TAG}_a -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/prof_${TAG}.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL --output-format csv -d $OUT/prof_${TAG}_b -- python3 $REPO/bench.py $ARGS > /dev/null 2>> $OUT/prof_${TAG}.err
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_${TAG}_a $OUT/prof_${TAG}_b > $OUT/${TAG}_summary.json
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete
echo done

#!/bin/bash
# rocprofv3 PMC passes over the cfg3 training step (instruction mix / LDS / wait counters of the fused backward kernel)
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
ARGS="--train --steps 2 --warmup 1 --no-cpu-baseline --no-grid-roofline --no-fat-schedule --no-fp16-leg --no-occupancy --no-dense192"
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/prof_tpmc_a -- python3 $REPO/bench.py $ARGS > $OUT/prof_tpmc_a.json 2> $OUT/prof_tpmc_a.err
echo "pass a done"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/prof_tpmc_b -- python3 $REPO/bench.py $ARGS > $OUT/prof_tpmc_b.json 2> $OUT/prof_tpmc_b.err
echo "pass b done"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/prof_tpmc_c -- python3 $REPO/bench.py $ARGS > $OUT/prof_tpmc_c.json 2> $OUT/prof_tpmc_c.err || echo "pass c failed"
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_tpmc_a $OUT/prof_tpmc_b $OUT/prof_tpmc_c > $OUT/train_pmc_summary.json
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete

#!/bin/bash
# rocprofv3 kernel trace of the cfg3 training step (bench.py --train) on the GPU box
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 $REPO/bench.py --train-only --steps 12 --warmup 2 > $OUT/prof_train_bench.json 2> $OUT/prof_train.err
cd $REPO
find $OUT/prof_train -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats.csv \;
find $OUT/prof_train -name "*.db" -delete
find $OUT/prof_train -name "*kernel_trace.csv" -delete
cut -c1-160 $OUT/train_kernel_stats.csv | head -40

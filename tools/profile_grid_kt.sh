#!/bin/bash
# rocprofv3 kernel trace + stats of the grid-encoder micro-benchmark (tools/grid_bench.py): per-kernel durations for DESIGN 4 / bench's
# roofline_gridencoder legs.  Summary -> gpurun_out/grid_kernel_stats.csv
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_grid_kt -- python3 $REPO/tools/grid_bench.py > $OUT/prof_grid_kt.log 2>&1
cd $REPO
find $OUT/prof_grid_kt -name "*kernel_stats.csv" -exec cp {} $OUT/grid_kernel_stats.csv \;
find $OUT/prof_grid_kt -name "*.db" -delete
find $OUT/prof_grid_kt -name "*kernel_trace.csv" -delete
cut -c1-170 $OUT/grid_kernel_stats.csv | head -20
cat $OUT/prof_grid_kt.log | tail -12

#!/bin/bash
# rocprofv3 kernel trace of the cfg3 training step (bench.py --train-only) on the GPU box; extra arguments go to bench.py,
# e.g. tools/profile_train.sh --train-records f16; TAG names the output (default train)
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
TAG=${TAG:-train}
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $REPO/bench.py --train-only --steps 12 --warmup 2 "$@" > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_$TAG.err
cd $REPO
# per-kernel median / mean without the warm-up steps (rocprofv3's own --stats table averages every launch, warm-ups included)
python3 tools/kernel_medians.py $OUT/prof_$TAG --skip 2 > $OUT/${TAG}_kernel_stats.csv
find $OUT/prof_$TAG -name "*.db" -delete
find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete
cut -c1-160 $OUT/${TAG}_kernel_stats.csv | head -40

set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg5 -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-side-legs --precision f16 --size 1024 --scene ellipsoid > $OUT/prof_cfg5_bench.json 2> $OUT/prof_cfg5.err
cd $REPO
find $OUT/prof_cfg5 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_cfg5.csv \;
find $OUT/prof_cfg5 -name "*.db" -delete; find $OUT/prof_cfg5 -name "*kernel_trace.csv" -delete
cut -c1-150 $OUT/kernel_stats_cfg5.csv | head -8

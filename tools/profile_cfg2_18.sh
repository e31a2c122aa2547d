#!/bin/bash
# kernel trace of the fused cfg2 frame under the reference schedule (1, 8): tools/profile_cfg2_18.sh [f32|f16] -> gpurun_out/cfg2_18_<prec>_kernel_stats.csv
export TMPDIR=/tmp
REPO=$(pwd)
PREC=${1:-f32}
OUT=$REPO/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg2_18_${PREC}_kt -- python3 $REPO/tools/cfg2_bench.py $PREC 1,8 > $OUT/cfg2_18_${PREC}_bench.json 2> $OUT/cfg2_18_${PREC}.err
cd $REPO
find $OUT/cfg2_18_${PREC}_kt -name "*kernel_stats.csv" -exec cp {} $OUT/cfg2_18_${PREC}_kernel_stats.csv \;
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/cfg2_18_${PREC}_kernel_stats.csv")))
frames = [int(r["Calls"]) for r in rows if "loop_begin" in r["Name"]][0]
for r in rows[:6]:
    print("%-46s %6.1f calls/frame %8.1f us/frame  avg %6.2f us" % (r["Name"][:46], int(r["Calls"]) / frames, float(r["TotalDurationNs"]) / frames / 1e3, float(r["AverageNs"]) / 1e3))
PY

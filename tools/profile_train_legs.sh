#!/bin/bash
# kernel trace of bench.py's four cfg3 training legs run in ONE process (as the default bench line does)
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trainlegs -- python3 $REPO/bench.py --no-cpu-baseline --no-grid-roofline --no-fat-schedule --no-fp16-leg --no-occupancy --no-dense192 --no-cfg5 --steps 5 > $OUT/prof_trainlegs_bench.json 2> $OUT/prof_trainlegs.err
cd $REPO
find $OUT/prof_trainlegs -name "*kernel_stats.csv" -exec cp {} $OUT/trainlegs_kernel_stats.csv \;
find $OUT/prof_trainlegs -name "*.db" -delete
find $OUT/prof_trainlegs -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, json
for r in list(csv.DictReader(open("$OUT/trainlegs_kernel_stats.csv")))[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3), "min %9.1f max %9.1f" % (float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
d = json.loads(open("$OUT/prof_trainlegs_bench.json").read().strip().splitlines()[-1])
for k in d:
    if k.startswith("train_step"): print(k, d[k]["ms_per_step"])
PY

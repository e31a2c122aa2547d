#!/bin/bash
# PMC passes (HBM-side traffic) over the grid-encoder micro-benchmark; summaries -> gpurun_out/grid_pmc_summary.json
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_grid_fetch -- python3 $REPO/tools/grid_bench.py > $OUT/prof_grid_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_grid_write -- python3 $REPO/tools/grid_bench.py > $OUT/prof_grid_write.log 2>&1
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_grid_fetch $OUT/prof_grid_write > $OUT/grid_pmc_summary.json
find $OUT/prof_grid_fetch $OUT/prof_grid_write -name "*.db" -delete
find $OUT/prof_grid_fetch $OUT/prof_grid_write -name "*counter_collection.csv" -delete
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/grid_pmc_summary.json"))
for c, ks in d.items():
    for k, v in ks.items():
        if "grid" in k:
            print(c, k[:70], v["launches"], v["avg_per_launch"], v["max"])
PY

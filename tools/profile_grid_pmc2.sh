#!/bin/bash
# cache-side counters of the grid-encoder micro-benchmark: how many L1 -> L2 read requests the gathers make (the forward's bound)
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
rocprofv3 -L > $OUT/rocprof_counters.txt 2>&1 || true
grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*" $OUT/rocprof_counters.txt | sort -u > $OUT/rocprof_cache_counters.txt || true
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/prof_grid_tcp -- python3 $REPO/tools/grid_bench.py > $OUT/prof_grid_tcp.log 2>&1 || echo "tcp pass failed"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_grid_tcc -- python3 $REPO/tools/grid_bench.py > $OUT/prof_grid_tcc.log 2>&1 || echo "tcc pass failed"
rocprofv3 --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum --output-format csv -d $OUT/prof_grid_ta -- python3 $REPO/tools/grid_bench.py > $OUT/prof_grid_ta.log 2>&1 || echo "ta pass failed"
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_grid_tcp $OUT/prof_grid_tcc $OUT/prof_grid_ta > $OUT/grid_cache_pmc_summary.json
find $OUT/prof_grid_tcp $OUT/prof_grid_tcc $OUT/prof_grid_ta -name "*.db" -delete 2>/dev/null || true
find $OUT/prof_grid_tcp $OUT/prof_grid_tcc $OUT/prof_grid_ta -name "*counter_collection.csv" -delete 2>/dev/null || true
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/grid_cache_pmc_summary.json"))
for c, ks in d.items():
    for k, v in ks.items():
        if "lmp<float" in k or "untile" in k or "forward_lds<float, 2u" in k:
            print(c, k[:60], v["launches"], v["avg_per_launch"], v["max"])
PY
wc -l $OUT/rocprof_cache_counters.txt; head -80 $OUT/rocprof_cache_counters.txt | tr '\n' ' '

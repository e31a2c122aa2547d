#!/bin/bash
# texture-addresser / L1 counters of the fused frame kernel (run through gpurun from the repo root):
#   tools/profile_frame_cache.sh [f32|f16] [extra bench.py args]
# answers "is the f16 frame kernel bound by the gather path (TA busy, L1 fills) or by vector issue?".  Separate --pmc passes, no trace
# domains next to them; no TCC_EA* counters (a pass with those never finished on this pool).
set -e
export TMPDIR=/tmp
REPO=$(pwd)
PREC=${1:-f16}
shift || true
ARGS="--steps 5 --warmup 2 --no-side-legs --no-clock-probe --precision $PREC $*"
OUT=$REPO/gpurun_out
P=$OUT/prof_fc_${PREC}
mkdir -p $OUT
cd /tmp
pass() {   # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d ${P}_$name -- python3 $REPO/bench.py $ARGS > ${P}_${name}_bench.json 2> ${P}_$name.err || echo "$name pass failed"
    echo "$name done"
}
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass ta3 TA_BUSY_avr TA_ADDR_STALLED_BY_TD_CYCLES_sum
pass tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp3 TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
cd $REPO
python3 tools/summarize_pmc.py ${P}_ta1 ${P}_ta2 ${P}_ta3 ${P}_tcp1 ${P}_tcp2 ${P}_tcp3 > $OUT/frame_cache_pmc_${PREC}.json
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete
python3 - <<PY
import json
d = json.load(open("gpurun_out/frame_cache_pmc_${PREC}.json"))
for c, ks in d.items():
    for k, v in ks.items():
        if "lz_k_frame<" in k:
            print(c, k, v["launches"], v["avg_per_launch"])
PY

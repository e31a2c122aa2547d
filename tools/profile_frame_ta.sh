#!/bin/bash
# texture-addresser / L1 counters of the fused frame kernel: tools/profile_frame_ta.sh [f16|f32] [extra bench args] -> gpurun_out/frame_ta_<prec>.json
export TMPDIR=/tmp
REPO=$(pwd)
PREC=${1:-f16}
shift
OUT=$REPO/gpurun_out
ARGS="--precision $PREC --no-side-legs --no-cpu-baseline --no-clock-probe --steps 6 --warmup 2 $@"
cd /tmp
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_ta1_$PREC -- python3 $REPO/bench.py $ARGS > $OUT/prof_ta1_$PREC.json 2> $OUT/prof_ta1_$PREC.err || echo "ta1 failed"
rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $OUT/prof_ta2_$PREC -- python3 $REPO/bench.py $ARGS > $OUT/prof_ta2_$PREC.json 2> $OUT/prof_ta2_$PREC.err || echo "ta2 failed"
rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_FLAT_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/prof_ta3_$PREC -- python3 $REPO/bench.py $ARGS > $OUT/prof_ta3_$PREC.json 2> $OUT/prof_ta3_$PREC.err || echo "ta3 failed"
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_ta1_$PREC $OUT/prof_ta2_$PREC $OUT/prof_ta3_$PREC > $OUT/frame_ta_$PREC.json
python3 - <<PY
import json
d = json.load(open("$OUT/frame_ta_$PREC.json"))
for c, ks in d.items():
    for k, v in ks.items():
        if k.startswith("lz_k_frame<"): print(c, k, v["avg_per_launch"])
PY
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete

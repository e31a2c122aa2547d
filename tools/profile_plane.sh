#!/bin/bash
# counters of the stand-alone triplane plane encoder (tools/plane_bench.py): tools/profile_plane.sh -> gpurun_out/plane_pmc.json
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/prof_pl1 -- python3 $REPO/tools/plane_bench.py > /dev/null 2> $OUT/prof_pl1.err || echo "pl1 failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/prof_pl2 -- python3 $REPO/tools/plane_bench.py > /dev/null 2> $OUT/prof_pl2.err || echo "pl2 failed"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pl3 -- python3 $REPO/tools/plane_bench.py > /dev/null 2> $OUT/prof_pl3.err || echo "pl3 failed"
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_pl1 $OUT/prof_pl2 $OUT/prof_pl3 > $OUT/plane_pmc.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/plane_pmc.json"))
for c, ks in d.items():
    for k, v in ks.items():
        if "grid_forward" in k: print(c, v["avg_per_launch"], v["launches"])
PY
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete

#!/bin/bash
# counters of the table scatter on both sample-row layouts: tools/profile_scatter.sh -> gpurun_out/scatter_pmc_{ray,step}.json
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
cd /tmp
for l in ray step; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/prof_sc1_$l -- python3 $REPO/tools/scatter_bench.py $l > /dev/null 2> $OUT/prof_sc1_$l.err || echo "sc1 $l failed"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/prof_sc2_$l -- python3 $REPO/tools/scatter_bench.py $l > /dev/null 2> $OUT/prof_sc2_$l.err || echo "sc2 $l failed"
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $OUT/prof_sc3_$l -- python3 $REPO/tools/scatter_bench.py $l > /dev/null 2> $OUT/prof_sc3_$l.err || echo "sc3 $l failed"
  cd $REPO; python3 tools/summarize_pmc.py $OUT/prof_sc1_$l $OUT/prof_sc2_$l $OUT/prof_sc3_$l > $OUT/scatter_pmc_$l.json; cd /tmp
done
cd $REPO
python3 - <<PY
import json
for l in ("ray", "step"):
    d = json.load(open("gpurun_out/scatter_pmc_%s.json" % l))
    for c, ks in d.items():
        for k, v in ks.items():
            if "grid_backward" in k: print(l, c, v["avg_per_launch"], v["launches"])
PY
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete

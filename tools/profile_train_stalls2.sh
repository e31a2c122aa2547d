#!/bin/bash
# issue / wait / LDS counters of the training step's kernels: tools/profile_train_stalls2.sh [bench args, e.g. --train-forward f16 --train-backward f16]
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
TAG=${TAG:-tstalls}
ARGS="--train-only --steps 4 --warmup 2 $*"
cd /tmp
pass() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_${TAG}_$name -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/prof_${TAG}_$name.err || echo "$name failed"; echo "$name done"; }
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES
pass b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
pass c SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd $REPO
python3 tools/summarize_pmc.py $OUT/prof_${TAG}_a $OUT/prof_${TAG}_b $OUT/prof_${TAG}_c > $OUT/${TAG}_summary.json
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_summary.json"))
ks = sorted({k for c in d.values() for k in c if k.startswith("lz_k_") or "head" in k}, key=lambda k: -d["SQ_WAVE_CYCLES"].get(k, {"sum": 0})["sum"])[:6]
for k in ks:
    g = lambda c: d[c].get(k, {}).get("avg_per_launch", 0.0)
    wc = g("SQ_WAVE_CYCLES") or 1.0
    cu = g("SQ_BUSY_CU_CYCLES") or 1.0
    print(k[:58].ljust(58), "active %.2f wait_inst %.2f wait_any %.2f | lds_busy %.2f valu_active/wave %.2f mfma_busy %.2f | valu %d lds %d rd %d wr %d (x1e6)" % (
        g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_LDS_IDX_ACTIVE") / cu, g("SQ_ACTIVE_INST_VALU") / wc,
        g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * cu), g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_VMEM_RD") / 1e6, g("SQ_INSTS_VMEM_WR") / 1e6))
PY

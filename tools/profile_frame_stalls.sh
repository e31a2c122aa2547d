#!/bin/bash
# where do the waves of the fused frame kernel wait?  tools/profile_frame_stalls.sh [f32|f16] [bench args]
set -e
export TMPDIR=/tmp
REPO=$(pwd)
PREC=${1:-f16}
shift || true
ARGS="--steps 5 --warmup 2 --no-side-legs --no-clock-probe --precision $PREC $*"
OUT=$REPO/gpurun_out
P=$OUT/prof_fs_${PREC}
cd /tmp
pass() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d ${P}_$name -- python3 $REPO/bench.py $ARGS > /dev/null 2> ${P}_$name.err || echo "$name pass failed"; echo "$name done"; }
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES
pass b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD
pass d SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INST_LEVEL_VMEM
cd $REPO
python3 tools/summarize_pmc.py ${P}_a ${P}_b ${P}_c ${P}_d > $OUT/frame_stalls_${PREC}.json
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete
python3 - <<PY
import json
d = json.load(open("gpurun_out/frame_stalls_${PREC}.json"))
for c, ks in d.items():
    for k, v in ks.items():
        if "lz_k_frame<" in k:
            print(c, k, v["avg_per_launch"])
PY

#!/bin/bash
# LDS bank conflicts and wave-cycle split of EVERY kernel of four bench commands (the -O training step, the f32-exact one, the f32 and f16
# frames): one rocprofv3 --pmc pass each -> a table per command (kernels with >= 1 % of the busy cycles)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out
cd /tmp
for t in "rc --train-only --steps 3 --warmup 2 --train-forward f16 --train-backward f16" "f32 --train-only --steps 3 --warmup 2" \
         "frame32 --steps 3 --warmup 2 --no-side-legs --no-cpu-baseline" "frame16 --steps 3 --warmup 2 --no-side-legs --no-cpu-baseline --precision f16"; do
  set -- $t; tag=$1; shift
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/prof_lds_$tag -- python3 $R/bench.py "$@" > /dev/null 2> $O/prof_lds_$tag.err
done
cd $R
python3 - <<PY
import json, subprocess, sys
for tag in ("rc", "f32", "frame32", "frame16"):
    d = json.loads(subprocess.run([sys.executable, "tools/summarize_pmc.py", "gpurun_out/prof_lds_" + tag], capture_output=True, text=True).stdout)
    print("==", tag)
    tot = sum(x["sum"] for x in d["SQ_BUSY_CU_CYCLES"].values())
    for k, v in d["SQ_BUSY_CU_CYCLES"].items():
        busy = v["sum"]
        if busy < 0.01 * tot:
            continue
        g = lambda c: d[c].get(k, {}).get("sum", 0.0)
        print("  %-60s busy %5.1f%%  lds_active/busy %.2f  conflict/lds_active %.2f  wave: active %.2f wait_any %.2f wait_inst %.2f" % (k[:60], 100 * busy / tot, g("SQ_LDS_IDX_ACTIVE") / busy, g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1)))
PY
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete

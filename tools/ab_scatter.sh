#!/bin/bash
# per-launch time of the table scatter inside the cfg3 training step for library variants: tools/ab_scatter.sh "base grp8 ..."
for v in ${1:-base}; do
  if [ "$v" = base ]; then unset LZZX_NERF_HIP_SO; else export LZZX_NERF_HIP_SO=$(pwd)/lzzx_nerf_amd/lib/variants/$v.so; fi
  TAG=ab_sc_$v bash tools/profile_train.sh --train-forward f16 --train-backward f16 > gpurun_out/ab_sc_$v.log 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ab_sc_${v}_kernel_stats.csv")):
    if "lds_fx" in r["Name"]: print("$v", "avg %.0f min %.0f max %.0f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done

#!/bin/bash
# average launch time of the kernels matching a pattern inside the cfg3 training step, per library variant: tools/ab_kernel.sh PATTERN "base v1 v2" [bench args]
PAT=$1; VARS=${2:-base}; shift; shift
for v in $VARS; do
  if [ "$v" = base ]; then unset LZZX_NERF_HIP_SO; else export LZZX_NERF_HIP_SO=$(pwd)/lzzx_nerf_amd/lib/variants/$v.so; fi
  TAG=abk_$v bash tools/profile_train.sh --train-forward f16 --train-backward f16 "$@" > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/abk_${v}_kernel_stats.csv")):
    if "$PAT" in r["Name"]: print("$v", r["Name"][:40], "avg %.1f min %.1f max %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done

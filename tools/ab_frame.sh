#!/bin/bash
# A/B of library variants (lzzx_nerf_amd/build.py --variant) on the fused frame: tools/ab_frame.sh "base prio1 prio2" "f16 f32" [bench args]
VARS=${1:-base}
PRECS=${2:-f16}
shift; shift
for v in $VARS; do
  for p in $PRECS; do
    if [ "$v" = base ]; then unset LZZX_NERF_HIP_SO; else export LZZX_NERF_HIP_SO=$(pwd)/lzzx_nerf_amd/lib/variants/$v.so; fi
    python3 bench.py --precision $p --no-side-legs --no-cpu-baseline --no-clock-probe --steps 20 --warmup 3 "$@" > gpurun_out/ab_${v}_${p}.json 2> gpurun_out/ab_${v}_${p}.err || { echo "$v $p FAILED"; tail -3 gpurun_out/ab_${v}_${p}.err; continue; }
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/ab_${v}_${p}.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("$v $p ms_per_step %.4f kernel_ms %s value %.4g" % (d["ms_per_step"], r.get("avg_launch_ms", r.get("kernel_ms")), d["value"]))
PY
  done
done

#!/bin/bash
# Request-rate roofline of the cfg2 gather: L1 -> L2 read requests of the level-major gather kernel per second against the
# chip's measured line-fill peak (tools/probes/l1_fill_probe.hip, the same counter over the probe tells the counter's unit per 128-byte fill).
#   tools/profile_grid_req.sh -> gpurun_out/grid_req_summary.json
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
hipcc --offload-arch=gfx950 -O3 $REPO/tools/probes/l1_fill_probe.hip -o /tmp/l1_fill_probe 2> /dev/null
/tmp/l1_fill_probe 2 49 512 > $OUT/l1_fill_probe.log
cat $OUT/l1_fill_probe.log
cd /tmp
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $OUT/prof_req_probe -- /tmp/l1_fill_probe 49 > /dev/null 2> $OUT/prof_req_probe.err || echo "probe pmc failed"
for c in march_f32 march_f16 ray_f32 ray_f16; do
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $OUT/prof_req_${c} -- python3 $REPO/tools/grid_order_bench.py $c > $OUT/prof_req_${c}.log 2>&1 || echo "$c pmc failed"
  echo "$c done"
done
cd $REPO
python3 - <<PY
import json, re, subprocess, sys
sys.path.insert(0, "tools")
out = {}
probe = open("gpurun_out/l1_fill_probe.log").read()
out["probe"] = [dict(mib=int(m.group(1)), ms_per_launch=float(m.group(2)), fills_per_s=float(m.group(3)), lane_loads_per_launch=float(m.group(4)))
                for m in re.finditer(r"buffer\s+(\d+) MiB.*?: ([\d.]+) ms per launch, ([\d.e+]+) line fills/s.*?([\d.e+]+) lane-loads per launch", probe)]
def summ(d):
    r = subprocess.run([sys.executable, "tools/summarize_pmc.py", d], capture_output=True, text=True)
    return json.loads(r.stdout) if r.stdout.strip() else {}
out["probe_pmc"] = summ("gpurun_out/prof_req_probe")
for c in ("march_f32", "march_f16", "ray_f32", "ray_f16"):
    out[c] = dict(pmc=summ("gpurun_out/prof_req_" + c), log=open("gpurun_out/prof_req_%s.log" % c).read()[-400:])
json.dump(out, open("gpurun_out/grid_req_summary.json", "w"), indent=1)
print(json.dumps({k: (v if k == "probe" else {c: {n[:40]: x["avg_per_launch"] for n, x in ks.items()} for c, ks in (v.get("pmc", v) if isinstance(v, dict) else {}).items()}) for k, v in out.items()}, indent=1)[:3000])
PY
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete

#!/bin/bash
# same box: the whole bench line (no CPU baseline) with a library variant and with this tree's library, leg by leg
#   tools/ab_bench_legs.sh <variant name under lzzx_nerf_amd/lib/variants>
V=$PWD/lzzx_nerf_amd/lib/variants/$1.so
for which in old new old new; do
  if [ $which = old ]; then export LZZX_NERF_HIP_SO=$V; else unset LZZX_NERF_HIP_SO; fi
  python3 bench.py --no-cpu-baseline --no-clock-probe --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
legs={'headline':d['ms_per_step']}
for k,v in d.items():
    if isinstance(v,dict) and 'ms_per_step' in v: legs[k]=v['ms_per_step']
th=d.get('talking_head_frame',{})
for p in ('f32','f16'):
    if p in th: legs['talking_'+p]=th[p]['ms_per_frame']
c2=d.get('cfg2_hashgrid_render',{})
for p in ('f32_tables','f16_tables'):
    if p in c2 and isinstance(c2[p].get('fused'), dict): legs['cfg2_'+p]=c2[p]['fused']['ms_per_frame']
print('$which', ' '.join('%s=%s'%(k,v) for k,v in legs.items()))
"
done

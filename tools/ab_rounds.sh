#!/bin/bash
# same box, alternating: the headline frame (fused, --cap per_ray so that both libraries take the same path) with THIS tree's library and with
# an earlier round's build of it (lzzx_nerf_amd/lib/variants/<name>.so, e.g. built from `git archive <commit> lzzx_nerf_amd/csrc include`)
#   tools/ab_rounds.sh r3_head [f16|f32]
V=lzzx_nerf_amd/lib/variants/$1.so
PREC=${2:-f16}
for i in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export LZZX_NERF_HIP_SO=$PWD/$V LZZX_NERF_HIP_SO_OLDER=1; else unset LZZX_NERF_HIP_SO LZZX_NERF_HIP_SO_OLDER; fi
    echo -n "$which $PREC: "
    python3 bench.py --no-side-legs --no-clock-probe --precision $PREC --steps 30 --warmup 5 --cap per_ray 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel', d['roofline'].get('avg_launch_ms'))"
  done
done

#!/bin/bash
# same box, alternating: the cfg3 training step legs with a library variant ("old") and this tree's library ("new")
#   tools/ab_train.sh <variant> ["bench args" ...]
V=$PWD/lzzx_nerf_amd/lib/variants/$1.so; shift
for i in 1 2 3; do for which in old new; do
  if [ $which = old ]; then export LZZX_NERF_HIP_SO=$V; else unset LZZX_NERF_HIP_SO; fi
  for a in "$@"; do
    python3 bench.py --train-only --steps 20 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['train_step']; print('$which', '[$a]', d['ms_per_step_median'], 'ms median', d['ms_per_step'], 'ms mean')"
  done
done; done

#!/bin/bash
# -O training step under library variants x sample-row layouts: tools/train_layout_ab.sh "base g16 ..." -> gpurun_out/r5_train_layout_ab.log
LOG=gpurun_out/r5_train_layout_ab.log
: > $LOG
for v in ${1:-base}; do
  if [ "$v" = base ]; then unset LZZX_NERF_HIP_SO; else export LZZX_NERF_HIP_SO=$(pwd)/lzzx_nerf_amd/lib/variants/$v.so; fi
  for l in ${2:-step}; do
    echo -n "$v $l: " >> $LOG
    timeout -k 10 200 python bench.py --train-only --train-forward f16 --train-backward f16 --train-layout $l --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['train_step']; print('ms_per_step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'loss', d['loss'])" >> $LOG || exit 1
  done
  timeout -k 10 120 python tools/scatter_bench.py step 2>/dev/null | tail -1 >> $LOG
done
cat $LOG

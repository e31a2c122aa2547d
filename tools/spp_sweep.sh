#!/bin/bash
# fused frame kernel: samples per ray and pass (S) against the ray count of a rank's tile (one GPU): tools/spp_sweep.sh [f32|f16] -> gpurun_out/r5_spp_<prec>.log
PREC=${1:-f16}
LOG=gpurun_out/r5_spp_$PREC.log
: > $LOG
for so in 2 4 8 16; do for spp in 1 2 4 8 16; do
  echo -n "shard_of=$so steps_per_pass=$spp: " >> $LOG
  timeout -k 10 120 python bench.py --no-side-legs --no-clock-probe --precision $PREC --steps 20 --warmup 5 --shard-of $so --steps-per-pass $spp 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'], 'rays', d['config']['rays_per_rank'])" >> $LOG || exit 1
done; done
cat $LOG

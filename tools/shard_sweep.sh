#!/bin/bash
# one rank's tile of a frame sharded 2 / 4 / 8 ways on ONE GPU (no collective): tools/shard_sweep.sh [f32|f16] -> gpurun_out/r5_shards_<prec>.log
PREC=${1:-f32}
LOG=gpurun_out/r5_shards_$PREC.log
: > $LOG
for so in 1 2 4 8; do for tl in interleaved contiguous; do
  if [ $so = 1 ] && [ $tl = contiguous ]; then continue; fi
  echo -n "shard_of=$so tiles=$tl: " >> $LOG
  timeout -k 10 120 python3 bench.py --no-side-legs --no-clock-probe --precision $PREC --steps 20 --warmup 5 --shard-of $so --tiles $tl 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'rays', d['config']['rays_per_rank'], 'samples', d['config']['samples_per_step'])" >> $LOG || exit 1
done; done
cat $LOG

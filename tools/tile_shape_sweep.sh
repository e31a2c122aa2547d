#!/bin/bash
# rank 0's tile of a frame sharded 8 / 4 ways (one GPU, no collective) under forced launch shapes: samples per pass S x slot rows
#   tools/tile_shape_sweep.sh [f32|f16] -> gpurun_out/r4_tile_shapes_<prec>.log
PREC=${1:-f16}
LOG=gpurun_out/r4_tile_shapes_$PREC.log
: > $LOG
for so in 8 4; do for S in 0 1 2 4; do for rows in 0 1 2; do
  if [ $PREC = f32 ] && [ $rows != 0 ]; then continue; fi
  echo -n "shard_of=$so S=$S rows=$rows: " >> $LOG
  if [ $rows = 0 ]; then unset LZ_FRAME_ROWS; else export LZ_FRAME_ROWS=$rows; fi
  timeout -k 10 120 python3 bench.py --no-side-legs --no-clock-probe --precision $PREC --steps 30 --warmup 5 --shard-of $so --tiles interleaved --steps-per-pass $S 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel', d['roofline'].get('avg_launch_ms'), 'samples', d['config']['samples_per_step'])" >> $LOG || exit 1
done; done; done
cat $LOG

#!/bin/bash
# rank 0's tile of an N-way sharded frame under forced samples-per-pass S: kernel time, samples, head rows (fill = samples / rows)
#   tools/tile_fill.sh [f32|f16] [ways] -> gpurun_out/tile_fill_<prec>.log
PREC=${1:-f16}; WAYS=${2:-8}
LOG=gpurun_out/tile_fill_$PREC.log
: > $LOG
for S in 0 1 2 4 8 16; do
  echo -n "shard_of=$WAYS S=$S: " >> $LOG
  timeout -k 10 120 python3 bench.py --no-side-legs --no-clock-probe --precision $PREC --steps 30 --warmup 5 --shard-of $WAYS --tiles interleaved --steps-per-pass $S 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['config']['samples_per_step']; print('ms_per_step', d['ms_per_step'], 'kernel', r.get('avg_launch_ms'), 'samples', s, 'rows', r.get('rows_per_frame'), 'fill %.3f' % (s / max(r.get('rows_per_frame') or 1, 1)))" >> $LOG || exit 1
done
cat $LOG

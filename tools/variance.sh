#!/bin/bash
# run-to-run spread of the headline numbers: five separate processes per configuration (fresh allocations, fresh clocks)
OUT=gpurun_out/r2_variance.log
: > $OUT
for cfg in "--precision f32" "--precision f16"; do
  for i in 1 2 3 4 5; do
    python bench.py --no-side-legs --steps 20 --warmup 5 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'])" >> $OUT || exit 1
  done
done
for cfg in "" "--train-forward f16 --train-backward f16"; do
  for i in 1 2 3 4 5; do
    python bench.py --train-only --steps 20 --warmup 3 $cfg 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin)['train_step']; print('train $cfg', d['ms_per_step'])" >> $OUT || exit 1
  done
done
cat $OUT

#!/bin/bash
# fused frame kernel: samples per ray and pass (S) against the ray count of a rank's tile (one GPU)
for so in 1 2 4 8 16; do for spp in 1 2 4 8; do
  echo "shard_of=$so steps_per_pass=$spp" >> gpurun_out/r2_spp.log
  timeout -k 10 120 python bench.py --no-side-legs --steps 20 --warmup 5 --shard-of $so --steps-per-pass $spp 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['rays_per_rank'], d['roofline']['rows_per_frame'], d['config']['samples_per_step'])" >> gpurun_out/r2_spp.log || exit 1
done; done

for so in 2 4 8; do for tl in interleaved contiguous; do for md in fused loop; do
  echo "shard_of=$so tiles=$tl mode=$md" >> gpurun_out/r2_shards.log
  timeout -k 10 120 python bench.py --no-side-legs --steps 20 --warmup 5 --shard-of $so --tiles $tl --mode $md 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['rays_per_rank'], d['config']['samples_per_step'])" >> gpurun_out/r2_shards.log || exit 1
done; done; done

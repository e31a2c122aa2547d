#!/bin/bash
# experiment: does the order rays enter the frame queue in (lane coherence of the gathers) move the fused frame kernel?
#   tools/exp_ray_order.sh [f16|f32]   needs variants km2 km4 km8 (build.py --variant kmN --only lz_frame.hip -- -DLZF_KEY_MERGE=N)
P=${1:-f16}
LOG=gpurun_out/r5_ray_order_$P.log
: > $LOG
for v in base km2 km4 km8; do
  for o in "" tile8x4 tile8x8 tile16x16 morton random; do
    [ "$v" != base ] && [ ! -f lzzx_nerf_amd/lib/variants/$v.so ] && continue
    export LZ_EXP_RAY_ORDER=$o
    echo -n "variant=$v order=${o:-rowmajor}: " >> $LOG
    tools/ab_frame.sh $v $P >> $LOG 2>&1 || exit 1
  done
done
cat $LOG

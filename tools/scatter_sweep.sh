#!/bin/bash
# tools/scatter_sweep.sh "base sp4 ..." -> gpurun_out/r5_scatter_sweep.log   (library variants of lz_grid.hip on both sample-row layouts)
LOG=gpurun_out/r5_scatter_sweep.log
: > $LOG
for v in ${1:-base}; do for l in ray step; do
  if [ "$v" = base ]; then unset LZZX_NERF_HIP_SO; else export LZZX_NERF_HIP_SO=$(pwd)/lzzx_nerf_amd/lib/variants/$v.so; fi
  echo -n "$v: " >> $LOG
  timeout -k 10 120 python tools/scatter_bench.py $l 2>/dev/null | tail -1 >> $LOG || exit 1
done; done
cat $LOG

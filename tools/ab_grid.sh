#!/bin/bash
# A/B of library variants on the grid-encoder micro-benchmark: tools/ab_grid.sh "base gld1 ..."
for v in ${1:-base}; do
  if [ "$v" = base ]; then unset LZZX_NERF_HIP_SO; else export LZZX_NERF_HIP_SO=$(pwd)/lzzx_nerf_amd/lib/variants/$v.so; fi
  echo "== $v"
  python3 tools/grid_bench.py 2>&1 | grep "cfg2 fwd.*layout=1\|cfg2 fwd.*layout=2" 
done

#!/bin/bash
# register / LDS / occupancy report of every kernel of one csrc file (clang's kernel-resource-usage remarks), e.g.
#   tools/kernel_resources.sh lz_head.hip
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Iinclude -c "lzzx_nerf_amd/csrc/$1" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|ScratchSize" | sed 's/^.*remark: //'

#!/bin/bash
# kernel-by-kernel timeline of the last bench frame (start offset, duration, gap to the previous kernel):
#   tools/frame_trace.sh [f32|f16] [extra bench.py args]      (through gpurun, from the repo root) -> gpurun_out/frame_trace_<prec>.txt
set -e
export TMPDIR=/tmp
REPO=$(pwd)
PREC=${1:-f16}
shift || true
OUT=$REPO/gpurun_out
D=$OUT/ftrace_${PREC}
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-side-legs --precision $PREC $* > ${D}_bench.json 2> ${D}.err
cd $REPO
python3 tools/frame_trace.py $D > $OUT/frame_trace_${PREC}.txt
find $D -name "*.db" -delete
find $D -name "*kernel_trace.csv" -delete
tail -40 $OUT/frame_trace_${PREC}.txt

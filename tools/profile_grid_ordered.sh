#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the ordered grid-encoder legs of bench.py, one process per case:
#   tools/profile_grid_ordered.sh -> gpurun_out/grid_ordered_pmc_summary.json  ({case: summarize_pmc output})
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out
cd /tmp
echo "{" > $OUT/grid_ordered_pmc_summary.json
first=1
for c in ray_f32 march_f32 ray_f16 march_f16; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_go_${c}_fetch -- python3 $REPO/tools/grid_order_bench.py $c > $OUT/prof_go_${c}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_go_${c}_write -- python3 $REPO/tools/grid_order_bench.py $c > $OUT/prof_go_${c}_write.log 2>&1
  if [ $first = 0 ]; then echo "," >> $OUT/grid_ordered_pmc_summary.json; fi
  first=0
  echo "\"$c\":" >> $OUT/grid_ordered_pmc_summary.json
  python3 $REPO/tools/summarize_pmc.py $OUT/prof_go_${c}_fetch $OUT/prof_go_${c}_write >> $OUT/grid_ordered_pmc_summary.json
  find $OUT/prof_go_${c}_fetch $OUT/prof_go_${c}_write -name "*.db" -delete
  find $OUT/prof_go_${c}_fetch $OUT/prof_go_${c}_write -name "*counter_collection.csv" -delete
  echo "$c done"
done
echo "}" >> $OUT/grid_ordered_pmc_summary.json
python3 -c "import json; d=json.load(open('$OUT/grid_ordered_pmc_summary.json')); print({c: {k: [(n[:30], v['avg_per_launch']) for n, v in ks.items() if 'grid' in n] for k, ks in s.items()} for c, s in d.items()})"

run() { echo "== $*"; timeout -k 10 400 python bench.py "$@" > gpurun_out/flag_run.json 2> gpurun_out/flag_run.err; rc=$?; python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/flag_run.json').read().strip().splitlines()[-1]); print('rc', $rc, 'value %.3e' % d['value'], 'ms', d['ms_per_step'], 'legs', [k for k in d if isinstance(d[k], dict)][:12], 'errors', d.get('leg_errors'))
except Exception as e:
    print('rc', $rc, 'NO JSON', e); print(open('gpurun_out/flag_run.err').read()[-600:])
"; }
run --size 256 --steps 5
run --size 1024 --scene ellipsoid --steps 5 --no-train --no-dense192
run --precision f16 --steps 5 --no-train
run --gather rgb24 --steps 5 --no-train --no-dense192 --no-grid-roofline --no-cpu-baseline
run --budget-factor 1 --n-step-cap 8 --steps 5 --no-train --no-dense192 --no-grid-roofline
run --train-mlp lz --steps 3 --no-dense192 --no-grid-roofline --no-cpu-baseline --no-fp16-leg --no-occupancy --no-reference-schedule
run --train-mlp torch --steps 3 --no-dense192 --no-grid-roofline --no-cpu-baseline --no-fp16-leg --no-occupancy --no-reference-schedule

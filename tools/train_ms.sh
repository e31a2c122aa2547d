#!/bin/bash
# ms per cfg3 training step for a list of arrangements: tools/train_ms.sh "<bench args>" "<bench args>" ...
for a in "$@"; do
  python3 bench.py --train-only --steps 10 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['train_step']; print('$a', '->', d['ms_per_step'], 'ms', d['samples_per_step'], 'samples')"
done

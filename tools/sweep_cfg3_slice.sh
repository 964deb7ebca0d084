#!/bin/bash
# cfg3 frame time for several evaluation-item lengths (PB_TUNE_SLICE); on the GPU box.
for s in 96 128 192 256 384 512; do
  PB_TUNE_SLICE=$s timeout -k 10 120 python bench.py --workload cfg3 --only --no-cpu-baseline --in-flight 0 --steps 50 --warmup 5 > /tmp/sw_$s.json || exit 1
  python -c "
import json,sys
d=json.loads(open('/tmp/sw_$s.json').read().strip().splitlines()[-1]); print('slice', $s, 'ms', round(d['ms_per_step'],4))
"
done

#!/bin/bash
# cfg1 frame time for several item lengths (PB_TUNE_SLICE) and XCD run lengths (PB_TUNE_XCDRUN); on the GPU box
for s in 32 48 64 96 128; do
  PB_TUNE_SLICE=$s timeout -k 10 120 python bench.py --workload cfg1 --only --no-cpu-baseline --in-flight 0 --steps 400 --warmup 40 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slice $s ms', round(d['ms_per_step'],5))" || exit 1
done
for x in 0 16 32 128; do
  PB_TUNE_XCDRUN=$x timeout -k 10 120 python bench.py --workload cfg1 --only --no-cpu-baseline --in-flight 0 --steps 400 --warmup 40 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcdrun $x ms', round(d['ms_per_step'],5))" || exit 1
done

#!/bin/bash
# cfg1 / cfg2 frame time for ray set-up blocks of 4, 8, 16 waves (PB_TUNE_RAYWAVES); on the GPU box
for w in cfg1 cfg2; do
  for n in 4 8 16; do
    PB_TUNE_RAYWAVES=$n timeout -k 10 120 python bench.py --workload $w --only --no-cpu-baseline --in-flight 0 --steps 400 --warmup 20 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w raywaves $n ms', round(d['ms_per_step'],5), d['frame_ok'])" || exit 1
  done
done

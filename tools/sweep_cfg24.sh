#!/bin/bash
# cfg2 and cfg4 frame time for several item lengths (PB_TUNE_SLICE) and XCD run lengths (PB_TUNE_XCDRUN); on the GPU box
run() { timeout -k 10 120 python bench.py --workload $1 --only --no-cpu-baseline --in-flight 0 --steps $2 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['ms_per_step'],5))" || exit 1; }
for s in 96 128 160 192; do PB_TUNE_SLICE=$s run cfg2 200 "cfg2 slice $s ms"; done
for x in 0 32 64 128; do PB_TUNE_XCDRUN=$x run cfg2 200 "cfg2 xcdrun $x ms"; done
for s in 256 384 512; do PB_TUNE_SLICE=$s run cfg4 30 "cfg4 slice $s ms"; done
for x in 0 64 256; do PB_TUNE_XCDRUN=$x run cfg4 30 "cfg4 xcdrun $x ms"; done

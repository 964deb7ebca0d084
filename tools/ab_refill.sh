#!/bin/bash
# frames with the bin lists kept (default) against frames that fill them again (PB_TUNE_REFILL); on the GPU box
for w in cfg1 cfg2 cfg3 cfg4; do
  steps=40; [ $w = cfg1 ] && steps=400; [ $w = cfg2 ] && steps=200
  for v in kept refill; do
    if [ $v = refill ]; then export PB_TUNE_REFILL=1; else unset PB_TUNE_REFILL; fi
    ms=$(timeout -k 10 100 python bench.py --workload $w --only --steps $steps --warmup 10 --no-cpu-baseline --in-flight 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['frame_ok'])") || exit 1
    echo "$w $v $ms"
  done
done

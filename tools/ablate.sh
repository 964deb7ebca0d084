#!/bin/bash
# Tuning aid: frame time with parts of the binned evaluation kernel switched off (results are then wrong).
for slice in ${SLICES:-512 128}; do
  for ab in ${ABLATE:-0 1 2 3 6 7}; do
    out=$(PB_TUNE_ABLATE=$ab PB_TUNE_SLICE=$slice PB_TUNE_WAVES=${WAVES:-8192} timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "slice=$slice ablate=$ab $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel_ms=%.4f" % d["roofline"]["kernel_ms"])')"
  done
done

#!/bin/bash
# Sweep the evaluation-queue tuning knobs on one GPU: prints kernel_ms per (slice, waves).
set -e
for slice in ${SLICES:-512 256 128 64}; do
  for waves in ${WAVES:-8192 4096 16384}; do
    out=$(PB_TUNE_SLICE=$slice PB_TUNE_WAVES=$waves timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1)
    echo "slice=$slice waves=$waves $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("kernel_ms=%.4f ms_per_step=%.4f" % (d["roofline"]["kernel_ms"], d["ms_per_step"]))')"
  done
done

#!/bin/bash
# eval_items_het2_kernel under the profiler (PB_DEBUG_FLAGS 8: no second pass); on the GPU box
for f in 0 8; do
  export PB_DEBUG_FLAGS=$f
  timeout -k 10 300 python tools/profile_frame.py cfg3 gpurun_out/s5_warm_$f trace > /dev/null || exit 1
  echo "flags $f: $(grep 'eval_items_het2' gpurun_out/s5_warm_$f/cfg3_kernel_stats.csv | cut -d, -f1,2,5 | head -1) $(grep residue_kernel gpurun_out/s5_warm_$f/cfg3_kernel_stats.csv | cut -d, -f1,4)"
done

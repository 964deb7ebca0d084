#!/bin/bash
# needs a library built with the timing experiments compiled in: make -C raytracer_amd/csrc clean all HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DPB_EXPERIMENTS"
# eval_items_het2_kernel time with parts of its work left out (PB_DEBUG_FLAGS: 8 no second pass, 16 no exact-order drains); on the GPU box
for f in 0 8 16 24; do
  export PB_DEBUG_FLAGS=$f
  timeout -k 10 300 python tools/profile_frame.py cfg3 gpurun_out/s4_h2exp_$f trace > /dev/null || exit 1
  echo "flags $f: $(grep 'eval_items_het2' gpurun_out/s4_h2exp_$f/cfg3_kernel_stats.csv | cut -d, -f1,2,5 | head -1) $(grep residue_kernel gpurun_out/s4_h2exp_$f/cfg3_kernel_stats.csv | cut -d, -f1,4)"
done

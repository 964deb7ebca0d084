#!/bin/bash
# needs a library built with the timing experiments compiled in: make -C raytracer_amd/csrc clean all HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DPB_EXPERIMENTS"
# residue_kernel time with parts of its work left out (PB_DEBUG_FLAGS: 1 no atomics, 2 no integral, 4 nothing); on the GPU box
for f in 0 1 2 4; do
  export PB_DEBUG_FLAGS=$f
  timeout -k 10 300 python tools/profile_frame.py cfg3 gpurun_out/s4_resexp_$f trace > /dev/null || exit 1
  echo "flags $f: $(grep residue_kernel gpurun_out/s4_resexp_$f/cfg3_kernel_stats.csv | cut -d, -f1,4)  $(grep 'eval_items_het2' gpurun_out/s4_resexp_$f/cfg3_kernel_stats.csv | cut -d, -f1,4 | head -1)"
done

#!/bin/bash
# needs a library built with the timing experiments compiled in: make -C raytracer_amd/csrc clean all HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DPB_EXPERIMENTS"
# eval_items_fast_kernel time with parts left out (PB_DEBUG_FLAGS: 128 no pair loop, 256 neither pair loop nor write-out); on the GPU box
for w in cfg1 cfg2; do for f in 0 128 256; do
  export PB_DEBUG_FLAGS=$f
  timeout -k 10 300 python tools/profile_frame.py $w gpurun_out/s4_evexp_$f trace > /dev/null || exit 1
  echo "$w flags $f: $(grep 'eval_items_fast_kernel<true>' gpurun_out/s4_evexp_$f/${w}_kernel_stats.csv | cut -d, -f1,2,4)"
done; done

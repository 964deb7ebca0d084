#!/bin/bash
# needs a library built with the timing experiments compiled in: make -C raytracer_amd/csrc clean all HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DPB_EXPERIMENTS"
# setup_fill_kernel time with one of its two roles left out (PB_DEBUG_FLAGS: 32 no ray set-up, 64 no binning); on the GPU box.
# PB_TUNE_REFILL: the kernel with both roles runs every frame (frames otherwise keep their lists and only set up rays).
export PB_TUNE_REFILL=1
for w in cfg1 cfg2; do for f in 0 32 64 96; do
  export PB_DEBUG_FLAGS=$f
  timeout -k 10 300 python tools/profile_frame.py $w gpurun_out/s4_setexp_$f trace > /dev/null || exit 1
  echo "$w flags $f: $(grep 'setup_fill' gpurun_out/s4_setexp_$f/${w}_kernel_stats.csv | cut -d, -f1,4)"
done; done

#!/bin/bash
# A/B on one box of what the first frames of a new segment set cost (tools/first_frame_cost.py, cfg3): the library built
# from other sources (raytracer_amd/lib/libphoton_beams_old.so, put there by hand) against the current one, alternating
#   bash tools/ab_first_frame.sh
L=raytracer_amd/lib
cp $L/libphoton_beams.so /tmp/new.so
for round in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp $L/libphoton_beams_old.so $L/libphoton_beams.so; else cp /tmp/new.so $L/libphoton_beams.so; fi
    echo $which; timeout -k 10 100 python tools/first_frame_cost.py cfg3 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
cp /tmp/new.so $L/libphoton_beams.so

#!/bin/bash
# A/B on one box: the library built from other sources (raytracer_amd/lib/libphoton_beams_old.so, put there by hand) against
# the current one, alternating, on the workloads named (default cfg3)
#   bash tools/ab_libs.sh [cfg1 cfg2 ...]
L=raytracer_amd/lib
W=${@:-cfg3}
cp $L/libphoton_beams.so /tmp/new.so
for round in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp $L/libphoton_beams_old.so $L/libphoton_beams.so; else cp /tmp/new.so $L/libphoton_beams.so; fi
    for w in $W; do
      steps=40; [ $w = cfg1 ] && steps=400; [ $w = cfg2 ] && steps=200
      ms=$(timeout -k 10 100 python bench.py --workload $w --only --steps $steps --warmup 10 --no-cpu-baseline --in-flight 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])") || exit 1
      echo "$which $w $ms"
    done
  done
done
cp /tmp/new.so $L/libphoton_beams.so

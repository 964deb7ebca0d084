#!/bin/bash
# usage: tools/sweep_env.sh <workload> VAR=v1,v2 [VAR2=...]   (cartesian product; prints ms per frame)
wl=$1; shift
run() { # env assignments in "$@"
  out=$(env "$@" timeout -k 10 120 python bench.py --only --no-cpu-baseline --in-flight 0 --workload $wl --steps 100 --warmup 10) || { echo "$* FAILED"; return 1; }
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(sys.argv[2], 'ms_per_step=%.4f kernel_ms=%.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" "$out" "$*"
}
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}; new=()
  for c in "${combos[@]}"; do for v in ${vals//,/ }; do new+=("$c $var=$v"); done; done
  combos=("${new[@]}")
done
for c in "${combos[@]}"; do run $c || exit 1; done

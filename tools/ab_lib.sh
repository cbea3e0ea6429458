#!/bin/bash
# Same-box comparison of two builds of the library: alternates bench runs of each.
# usage: tools/ab_lib.sh <base.so> [bench.py args...]
base=$1; shift
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export MUZERO_AMD_LIB=$PWD/$base; else unset MUZERO_AMD_LIB; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['value']), 'sims/s', round(d['ms_per_step'],3), 'ms  chain', round(d['roofline']['launch_us'],1), 'us')" || exit 1
  done
done

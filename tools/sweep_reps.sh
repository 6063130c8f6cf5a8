#!/bin/bash
# usage: tools/sweep_reps.sh <flags> "<reps> <reps> ..."  -> config-2 shape with a different number of chains
for r in $2; do
  echo -n "flags=$1 reps=$r: "
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --steps 8 --warmup 2 --debug-flags $1 --reps $r 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch  %.3e steps/s  window %d launches %d mode %s' % (d['launch_ms']['mean'], d['value'], d['config']['window'], d['config']['launches'], d['config']['exec_mode']))"
done

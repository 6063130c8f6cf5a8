#!/bin/bash
# usage: tools/sweep_rpw.sh "<reps...>" "<rows per producer wave...>"  -> config-2 shape, launch time per combination (0 = automatic)
for r in $1; do
  for w in $2; do
    f=$((w << 8))
    echo -n "reps=$r rows/wave=$w: "
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --steps 8 --warmup 2 --debug-flags $f --reps $r 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch  window %d launches %d' % (d['launch_ms']['mean'], d['config']['window'], d['config']['launches']))"
  done
done

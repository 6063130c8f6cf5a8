#!/bin/bash
# usage: tools/ab_wave.sh "<reps...>"  -> wavefront-per-chain mode on the previous build and on the current one
for r in $1; do
  for lib in prev hip; do
    echo -n "$lib wave mode reps=$r: "
    MCSAS_HIP_LIB=$PWD/mcsas_amd/lib/libmcsas_$lib.so timeout -k 10 300 python bench.py --reps $r --mode 1 --mc-steps 2000 --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch (min %.3f)  %.3e steps/s' % (d['launch_ms']['mean'], d['launch_ms']['min'], d['value']))"
  done
done

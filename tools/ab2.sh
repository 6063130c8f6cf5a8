#!/bin/bash
# usage: tools/ab2.sh [rounds]  -> config-2 launch time of lib/libmcsas_prev.so (the last committed state) and of the current build, alternating, same box
for i in $(seq 1 ${1:-3}); do
  for lib in prev hip; do
    echo -n "$lib: "
    MCSAS_HIP_LIB=$PWD/mcsas_amd/lib/libmcsas_$lib.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch (min %.3f)  %.3e steps/s' % (d['launch_ms']['mean'], d['launch_ms']['min'], d['value']))"
  done
done

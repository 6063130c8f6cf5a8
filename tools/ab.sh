#!/bin/bash
# usage: tools/ab.sh "<flags...>" [reps]  -> each flag value on the previous build (lib/libmcsas_prev.so) and on the current one, same box
for f in $1; do
  for lib in prev hip; do
    echo -n "$lib flags=$f reps=${2:-50}: "
    MCSAS_HIP_LIB=$PWD/mcsas_amd/lib/libmcsas_$lib.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --steps 10 --warmup 3 --debug-flags $f --reps ${2:-50} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch (min %.3f)  %.3e steps/s  chi2 %.4f' % (d['launch_ms']['mean'], d['launch_ms']['min'], d['value'], d['final_chisq_median']))"
  done
done

#!/bin/bash
# usage: tools/sweep_flags.sh "<flags> <flags> ..."   -> one line per --debug-flags value (config 2 shape)
mkdir -p gpurun_out
for f in $1; do
  echo -n "flags=$f (rpw=$((f>>8)) bits=$((f&255))): "
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --steps 8 --warmup 2 --debug-flags $f 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch  %.3e steps/s  window %d launches %d chi2 %.4f' % (d['launch_ms']['mean'], d['value'], d['config']['window'], d['config']['launches'], d['final_chisq_median']))"
done

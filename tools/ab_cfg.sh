#!/bin/bash
# usage: tools/ab_cfg.sh "<config numbers>"  -> bench.py --config N on the previous build (lib/libmcsas_prev.so) and on the current one
for c in $1; do
  for lib in prev hip; do
    echo -n "$lib config=$c: "
    MCSAS_HIP_LIB=$PWD/mcsas_amd/lib/libmcsas_$lib.so timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-convergence-run --no-configs --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch (min %.3f)  %.3e steps/s' % (d['launch_ms']['mean'], d['launch_ms']['min'], d['value']))"
  done
done

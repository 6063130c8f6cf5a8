#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3_t7.log 2>&1; tail -4 gpurun_out/r3_t7.log
for c in 3 5 4; do
  echo -n "config $c: "
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-convergence-run --no-configs --steps 6 --warmup 2 --mc-steps 5000 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch (min %.3f)  %.3e steps/s window %d' % (d['launch_ms']['mean'], d['launch_ms']['min'], d['value'], d['config']['window']))"
done

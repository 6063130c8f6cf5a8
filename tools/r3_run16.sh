#!/bin/bash
for inf in 1 2 1 2; do
  echo -n "inflight=$inf: "
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-convergence-run --no-configs --inflight $inf --steps 10 --warmup 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4e  ms_per_step %.2f  launch median %.3f ms  (wall per launch %.3f ms)' % (d['value'], d['ms_per_step'], d['launch_ms']['median'], d['ms_per_step'] / 40))"
done

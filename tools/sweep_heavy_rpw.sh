#!/bin/bash
# rows per producer wave (tuning word bits 8-11) for the configurations whose rows cost an integral: MC steps/s of one analysis at a time
cd ${GRAFT_REPO_ROOT:-$PWD}
for cfg in 3 5; do
  for rpw in 0 1 2 3 4; do
    f=$((rpw << 8))
    v=$(python3 bench.py --config $cfg --scaling weak --steps 3 --warmup 1 --launches-per-step 2 --mc-steps 5000 --debug-flags $f --no-cpu-baseline --no-convergence-run --no-configs --no-series 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4g steps/s, launch %.2f ms, window %d' % (d['value'], d['launch_ms']['median'], d['config']['window']))")
    echo "config $cfg rows/wave request $rpw: $v"
  done
done

#!/bin/bash
# execution modes for the configurations whose rows cost an integral, at their TOTAL repetition counts on one GPU (strong scaling, N = 1)
cd ${GRAFT_REPO_ROOT:-$PWD}
for spec in "3 200 3000" "3 64 3000" "3 400 3000" "5 100 3000" "5 300 2000" "4 400 600" "4 100 1000"; do
  set -- $spec
  for mode in 0 1 2 3; do
    v=$(python3 bench.py --config $1 --scaling weak --reps $2 --mode $mode --steps 2 --warmup 1 --launches-per-step 1 --mc-steps $3 --no-cpu-baseline --no-convergence-run --no-configs --no-series 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4g steps/s, launch %.1f ms, %s window %d' % (d['value'], d['launch_ms']['median'], d['config']['exec_mode'], d['config']['window']))" 2>/dev/null || echo "refused")
    echo "config $1 reps $2 mode $mode: $v"
  done
done

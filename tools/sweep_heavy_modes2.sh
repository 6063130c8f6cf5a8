#!/bin/bash
# where the wavefront mode overtakes the pipeline for rows with an integral: many chains, short budgets
cd ${GRAFT_REPO_ROOT:-$PWD}
IFS=";" read -ra LIST <<< "${SPECS:-3 1024 500;3 2048 500;3 4096 300;5 1024 500;5 2048 500;4 1024 200;4 2048 200}"
for spec in "${LIST[@]}"; do
  IFS=" " read -r a b c <<< "$spec"; set -- $a $b $c
  for mode in 1 2 3; do
    v=$(python3 bench.py --config $1 --scaling weak --reps $2 --mode $mode --steps 1 --warmup 1 --launches-per-step 1 --mc-steps $3 --no-cpu-baseline --no-convergence-run --no-configs --no-series 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4g steps/s, launch %.1f ms, %s window %d' % (d['value'], d['launch_ms']['median'], d['config']['exec_mode'], d['config']['window']))" 2>/dev/null || echo "refused")
    echo "config $1 reps $2 mode $mode: $v"
  done
done

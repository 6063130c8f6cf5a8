#!/bin/bash
# window length of the row-queue pipeline (rows with an integral), measurement knob MCSAS_HIP_PIPE_KB:
#   SPECS="<config>:<kb>[:<reps>] ..." tools/sweep_heavy_kb.sh
for spec in ${SPECS:-4:256 4:240 4:200}; do
  IFS=: read cfg kb reps <<< "$spec"
  MCSAS_HIP_PIPE_KB=$kb timeout -k 5 200 python3 bench.py --config $cfg ${reps:+--reps $reps} --scaling weak --steps 3 --warmup 1 --launches-per-step 2 --mc-steps ${STEPS:-5000} --no-cpu-baseline --no-convergence-run --no-configs --no-series 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(\"config\", d[\"config\"][\"baseline_config\"], \"reps\", d[\"config\"][\"reps_per_gpu\"], \"kb $kb: %.4g steps/s, launch %.2f ms, window %d\" % (d[\"value\"], d[\"launch_ms\"][\"median\"], d[\"config\"][\"window\"]))"
done

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_t6.log 2>&1; tail -6 gpurun_out/r3_t6.log
(time timeout -k 10 600 python bench.py) > gpurun_out/r3_bench1.log 2> gpurun_out/r3_bench1.err; tail -c 3000 gpurun_out/r3_bench1.log; tail -5 gpurun_out/r3_bench1.err
timeout -k 10 900 python tools/mode_sweep_r3.py > gpurun_out/r3_mode_sweep.jsonl 2> gpurun_out/r3_mode_sweep.err; wc -l gpurun_out/r3_mode_sweep.jsonl

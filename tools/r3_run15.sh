#!/bin/bash
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "edge_shapes or wide_q or 16384 or plugin" > gpurun_out/r3_t_wide.log 2>&1; rc=$?; tail -12 gpurun_out/r3_t_wide.log
[ $rc = 0 ] || exit $rc
timeout -k 10 600 python tools/wide_q_bench.py 1000 > gpurun_out/r3_wide_bench.log 2>&1; cat gpurun_out/r3_wide_bench.log

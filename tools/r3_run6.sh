#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3_t9.log 2>&1; tail -4 gpurun_out/r3_t9.log
bash tools/ab1.sh "0 0 0" > gpurun_out/r3_ab9.log 2>&1; cat gpurun_out/r3_ab9.log

#!/bin/bash
# round-3 GPU run: parity, release bench, timeline
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3_t5.log 2>&1; tail -5 gpurun_out/r3_t5.log
bash tools/ab1.sh "0 0" > gpurun_out/r3_ab5.log 2>&1
cat gpurun_out/r3_ab5.log
tools/r3_tl.sh v5

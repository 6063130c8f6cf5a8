#!/bin/bash
# round-3 GPU run: parity, release bench, tuning variants
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_t4.log 2>&1; tail -5 gpurun_out/r3_t4.log
bash tools/ab1.sh "0 0 1572864 1048576 262144 786432 65536" > gpurun_out/r3_ab4.log 2>&1
cat gpurun_out/r3_ab4.log

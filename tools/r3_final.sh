#!/bin/bash
# final pass of a round: GPU tests, profiles, the default bench line.  Run on the GPU box from the repo root.
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3_final_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r3_final_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 900 bash tools/profile_r03.sh > gpurun_out/r3_final_prof.log 2>&1 || exit 1
tail -5 gpurun_out/r3_final_prof.log
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err || exit 1
tail -c 3000 gpurun_out/r3_final_bench.json

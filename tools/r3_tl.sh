#!/bin/bash
# stamps-build timeline of one tick of config 2: tools/r3_tl.sh <tag> [debug flags]
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
MCSAS_DEBUG_FLAGS=${2:-0} MCSAS_HIP_LIB=$PWD/mcsas_amd/lib/libmcsas_stamps.so MCSAS_TIMELINE_TICK=${MCSAS_TIMELINE_TICK:-40} timeout -k 10 120 python tools/pipeline_stamps.py > gpurun_out/r3_tl_$1.log 2>&1
python tools/timeline_summary.py gpurun_out/r3_tl_$1.log 50 2>&1 | head -4
grep "mcsas timeline\] 6[0-1] " gpurun_out/r3_tl_$1.log | head -16
grep "mcsas timeline\] [0-1] " gpurun_out/r3_tl_$1.log | head -4
grep "stamps\] rep 0" gpurun_out/r3_tl_$1.log | tail -1
tail -1 gpurun_out/r3_tl_$1.log

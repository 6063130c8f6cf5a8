#!/bin/bash
# round-3 GPU run: parity of the overlapped producer, tuning variants, stamps timeline
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_t3.log 2>&1; tail -5 gpurun_out/r3_t3.log
bash tools/ab1.sh "0 524288 1048576 262144 65536 1024 2048 768 0" > gpurun_out/r3_ab3.log 2>&1
cat gpurun_out/r3_ab3.log
MCSAS_HIP_LIB=$PWD/mcsas_amd/lib/libmcsas_stamps.so MCSAS_TIMELINE_TICK=40 timeout -k 10 120 python tools/pipeline_stamps.py > gpurun_out/r3_tl3.log 2>&1
python tools/timeline_summary.py gpurun_out/r3_tl3.log 50 2>&1 | head -4
grep "mcsas timeline\] 6[0-1] " gpurun_out/r3_tl3.log | head -16
grep "stamps\] rep 0" gpurun_out/r3_tl3.log | tail -1

#!/bin/bash
L=mcsas_amd/lib
: > gpurun_out/r3_ka.log
for v in 0 1 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> gpurun_out/r3_ka.log
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python tools/placement_probe.py $L/libmcsas_hip.so 3 50 2>&1 | grep "^plan" >> gpurun_out/r3_ka.log || exit 1
done
echo "== unset" >> gpurun_out/r3_ka.log
timeout -k 10 200 python tools/placement_probe.py $L/libmcsas_hip.so 3 50 2>&1 | grep "^plan" >> gpurun_out/r3_ka.log
cat gpurun_out/r3_ka.log

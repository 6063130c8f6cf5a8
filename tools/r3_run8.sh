#!/bin/bash
L=mcsas_amd/lib
timeout -k 10 300 python tools/placement_probe.py $L/libmcsas_v1.so 6 60 > gpurun_out/r3_place.log 2>&1 && timeout -k 10 300 python tools/placement_probe.py $L/libmcsas_v1.so 6 60 >> gpurun_out/r3_place.log 2>&1
cat gpurun_out/r3_place.log

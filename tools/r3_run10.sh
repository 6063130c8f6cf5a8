#!/bin/bash
L=mcsas_amd/lib
: > gpurun_out/r3_place2.log
for i in 1 2 3 4 5 6 7 8 9 10; do
  echo "== process $i" >> gpurun_out/r3_place2.log
  timeout -k 10 100 python tools/placement_probe.py $L/libmcsas_v1.so 3 40 >> gpurun_out/r3_place2.log 2>&1 || exit 1
done
grep -c . gpurun_out/r3_place2.log
grep "^plan\|==" gpurun_out/r3_place2.log

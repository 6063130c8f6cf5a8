#!/bin/bash
L=mcsas_amd/lib
: > gpurun_out/r3_ab11.log
for pair in "prev hip" "hip prev" "prev c66" "prev v1"; do
  set -- $pair
  echo "== $1 vs $2" >> gpurun_out/r3_ab11.log; timeout -k 10 200 python tools/ab_pair.py $L/libmcsas_$1.so $L/libmcsas_$2.so 60 3 2>&1 | tail -3 >> gpurun_out/r3_ab11.log || exit 1
done
cat gpurun_out/r3_ab11.log

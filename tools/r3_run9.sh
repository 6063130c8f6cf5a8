#!/bin/bash
L=mcsas_amd/lib
: > gpurun_out/r3_ab9.log
for pair in "e84 c66" "c66 e84" "prev c66" "c66 prev" "prev v1" "v1 prev" "e84 c66"; do
  set -- $pair
  echo "== $1 vs $2" >> gpurun_out/r3_ab9.log; timeout -k 10 200 python tools/ab_pair.py $L/libmcsas_$1.so $L/libmcsas_$2.so 100 2>&1 | tail -3 >> gpurun_out/r3_ab9.log || exit 1
done
cat gpurun_out/r3_ab9.log

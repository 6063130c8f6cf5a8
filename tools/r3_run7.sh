#!/bin/bash
L=mcsas_amd/lib
for b in v1 e84 c66 v1rg8 v1rg2; do
  echo "== prev vs $b"; timeout -k 10 200 python tools/ab_pair.py $L/libmcsas_prev.so $L/libmcsas_$b.so 120 2>&1 | tail -3 || exit 1
done > gpurun_out/r3_ab7.log 2>&1
echo "== e84 vs c66" >> gpurun_out/r3_ab7.log; timeout -k 10 200 python tools/ab_pair.py $L/libmcsas_e84.so $L/libmcsas_c66.so 120 2>&1 | tail -3 >> gpurun_out/r3_ab7.log
cat gpurun_out/r3_ab7.log

#!/bin/bash
bash tools/ab1.sh "0 16 64 32 48 0" > gpurun_out/r3_abl.log 2>&1; cat gpurun_out/r3_abl.log

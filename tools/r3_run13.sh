#!/bin/bash
bash tools/r3_tl.sh e1 > gpurun_out/r3_tl_e1.out 2>&1; cat gpurun_out/r3_tl_e1.out | cut -c1-400

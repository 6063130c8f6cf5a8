#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
SWEEP_N=200,300,400,1000 SWEEP_R=64,96,128,160,200,256,320,400,512,640,800,1000,1500,2000,3000 timeout -k 10 1100 python tools/mode_sweep_r3.py > gpurun_out/r3_mode_sweep_dense.jsonl 2> gpurun_out/r3_mode_sweep_dense.err; wc -l gpurun_out/r3_mode_sweep_dense.jsonl; tail -2 gpurun_out/r3_mode_sweep_dense.err

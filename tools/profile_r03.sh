#!/bin/bash
# Round-3 profile passes, run ON THE GPU BOX from the repo root:   tools/profile_r03.sh
# One rocprofv3 run per counter group (no trace domain beside --kernel-trace in counter runs), the program directly
# after `--`.  Raw outputs under gpurun_out/r03prof/; tools/pmc_summary_r03.py turns them into profiles/r03_*.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r03prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --streams 1 --inflight 1"   # one analysis at a time: the kernels by themselves
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM"
run() { # tag, extra bench args
  tag=$1; shift
  echo "== $tag: $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_stats -- $B "$@" > $OUT/${tag}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/${tag}_sq1 -- $B "$@" > $OUT/${tag}_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $OUT/${tag}_sq2 -- $B "$@" > $OUT/${tag}_sq2.log 2>&1
  tail -1 $OUT/${tag}_stats.log | cut -c1-200
}
traffic() { # tag, extra bench args: memory-side bytes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_fetch -- $B "$@" > $OUT/${tag}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_write -- $B "$@" > $OUT/${tag}_write.log 2>&1
}
run c2 --steps 4 --warmup 1
traffic c2 --steps 2 --warmup 1
run c3 --config 3 --steps 2 --warmup 1 --mc-steps 5000
traffic c3 --config 3 --steps 2 --warmup 1 --mc-steps 5000
run c4 --config 4 --steps 2 --warmup 1 --mc-steps 2500
traffic c4 --config 4 --steps 2 --warmup 1 --mc-steps 2500
run c5 --config 5 --steps 2 --warmup 1 --mc-steps 5000
traffic c5 --config 5 --steps 2 --warmup 1 --mc-steps 5000
run wave8192 --reps 8192 --mode 1 --mc-steps 2000 --steps 2 --warmup 1
ls $OUT | head -40

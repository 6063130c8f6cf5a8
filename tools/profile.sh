#!/bin/bash
# The profile passes behind profiles/<round>_*, run ON THE GPU BOX from the repo root:   tools/profile.sh r04 [what ...]
#   what: c2 c3 c4 c5 wave8192 trace   (default: all)
# One rocprofv3 run per counter group (no trace domain beside --kernel-trace in counter runs), the program directly
# after `--`.  Raw outputs under gpurun_out/<round>prof/; tools/pmc_summary.py <round> [commit] turns them into profiles/<round>_*,
# tools/tick_trace.py turns the kernel trace into profiles/<round>_tick_trace.txt.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
ROUND=${1:-r04}; shift
WHAT=${*:-c2 c3 c4 c5 wave8192 trace}
OUT=$ROOT/gpurun_out/${ROUND}prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-convergence-run --no-configs --no-series --no-many-chains --scaling weak --launches-per-step 1 --streams 1 --inflight 1"   # one analysis at a time: the kernels by themselves
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
for w in $WHAT; do case $w in
  c2) run c2 --steps 4 --warmup 1; traffic c2 --steps 2 --warmup 1;;
  c3) run c3 --config 3 --steps 2 --warmup 1 --mc-steps 5000; traffic c3 --config 3 --steps 2 --warmup 1 --mc-steps 5000;;
  c4) run c4 --config 4 --steps 2 --warmup 1 --mc-steps 2500; traffic c4 --config 4 --steps 2 --warmup 1 --mc-steps 2500;;
  c5) run c5 --config 5 --steps 2 --warmup 1 --mc-steps 5000; traffic c5 --config 5 --steps 2 --warmup 1 --mc-steps 5000;;
  wave8192) run wave8192 --reps 8192 --mode 1 --mc-steps 20000 --steps 2 --warmup 1;;
  trace) rocprofv3 --kernel-trace --output-format csv -d $OUT/c2_trace -- $B --steps 6 --warmup 2 > $OUT/c2_trace.log 2>&1
         python3 $ROOT/tools/tick_trace.py $OUT/c2_trace > $ROOT/gpurun_out/${ROUND}_tick_trace.txt 2>&1; tail -3 $ROOT/gpurun_out/${ROUND}_tick_trace.txt | cut -c1-200;;
esac; done
ls $OUT | head -40

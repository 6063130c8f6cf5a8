#!/bin/bash
# per-tick kernel durations of config 2 (rocprofv3 kernel trace of a short bench run): tools/tick_trace.sh <tag> [lib.so]   (GPU box, repo root)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/ticktrace_$1
mkdir -p $OUT
[ -n "$2" ] && export MCSAS_HIP_LIB=$ROOT/$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --no-convergence-run --no-configs --no-series --no-many-chains --launches-per-step 1 --steps 6 --warmup 1 --inflight 1 > $OUT/run.log 2>&1
cd $ROOT && python3 tools/tick_trace.py $OUT

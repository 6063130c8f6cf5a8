#!/bin/bash
# usage (on the GPU box): tools/pmc_ab.sh "<debug flags...>"  -> SQ counter totals of pipe_tick_kernel per flag value
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_F64 SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_IFETCH"
for f in $1; do
  for g in 1 2; do
    eval "S=\$SQ$g"
    rocprofv3 --kernel-trace --pmc $S --output-format csv -d $OUT/f${f}_sq$g -- python3 $ROOT/bench.py --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --steps 2 --warmup 1 --debug-flags $f > $OUT/f${f}_sq$g.log 2>&1
    python3 - $OUT/f${f}_sq$g $f <<'PY'
import sys, glob, csv, collections
d, f = sys.argv[1], sys.argv[2]
tot = collections.Counter(); nd = set()
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "pipe_tick_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd.add(r["Dispatch_Id"])
print("flags", f, "dispatches", len(nd), " ".join("%s=%.4g" % (k, v / max(len(nd), 1)) for k, v in sorted(tot.items())))
PY
  done
done

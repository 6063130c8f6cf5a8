#!/usr/bin/env python3
"""One config-2 launch (chains x 20000 steps, pipeline mode) for diagnostics.  With the stamps
build of the library it prints the scan block's and the first producer block's phase shares to stderr:

    make -C mcsas_amd/csrc EXTRA=-DMCSAS_STAMPS OUT=../lib/libmcsas_hip_stamps.so BUILD=../../build/csrc_stamps
    MCSAS_HIP_LIB=$PWD/mcsas_amd/lib/libmcsas_hip_stamps.so MCSAS_REPS=50 MCSAS_DEBUG_FLAGS=0 python tools/pipeline_stamps.py

Under `rocprofv3 --kernel-trace` its trace feeds tools/trace_gaps.py (tick durations and gaps)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere()
m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
st = engine.Settings(n_contrib=400, n_reps=int(os.environ.get("MCSAS_REPS", "50")), max_iter=20000, conv_crit=0.0,
                     max_retries=0, seed=2, exec_mode=0, debug_flags=int(os.environ.get("MCSAS_DEBUG_FLAGS", "0")))
plan = engine.Plan(m.setup(), q, I, sig, st)
plan.launch(); res = plan.fetch()
plan.launch(); res = plan.fetch()
print("ms", plan.last_ms, "moves mean", res.num_moves.mean(), "iters", res.num_iter.mean(), plan.info)

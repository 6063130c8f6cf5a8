#!/usr/bin/env python3
"""K plans of ONE build on config 2 in one process, launches interleaved: does a plan's speed depend on where its buffers
landed?  usage: tools/placement_probe.py lib.so [plans] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["MCSAS_HIP_LIB"] = os.path.abspath(sys.argv[1])
os.environ["MCSAS_DEBUG_PTRS"] = "1"
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
st = engine.Settings(n_contrib=400, n_reps=50, max_iter=20000, conv_crit=0.0, max_retries=0, seed=20250101)
plans = [engine.Plan(m.setup(), q, I, sig, st) for _ in range(K)]
ms = [[] for _ in plans]
for i in range(n + 5):
    for k, pl in enumerate(plans):
        pl.reseed(1000 + i, 0); pl.launch(); pl.fetch(want_arrays=False)
        if i >= 5: ms[k].append(pl.last_ms)
for k in range(K):
    a = np.array(ms[k]); print("plan %d: median %.4f ms  min %.4f  max %.4f" % (k, np.median(a), a.min(), a.max()))

#!/usr/bin/env python3
"""Launch time of config 2 (50 chains x 20000 steps, pipeline) in THIS process's environment: tools/time_c2.py [launches] [plans]
(environment knobs of the library — MCSAS_HIP_UNCACHED, MCSAS_HIP_LIB — are read once per process: compare processes on one box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
st = engine.Settings(n_contrib=400, n_reps=int(os.environ.get("TIME_REPS", "50")), max_iter=20000, conv_crit=0.0, max_retries=0, seed=20250101,
                     debug_flags=int(os.environ.get("MCSAS_DEBUG_FLAGS", "0")))   # (non-zero: the tuning build's ablation word)
plans = [engine.Plan(m.setup(), q, I, sig, st) for _ in range(K)]
ms = [[] for _ in plans]
for i in range(n + 8):
    for k, pl in enumerate(plans):
        pl.reseed(1000 + i, 0); pl.launch()
        try:
            res = pl.fetch(want_arrays=(i == 0))
        except mcsas_amd._lib.McSASHipError as e:             # (diagnostic builds whose chains do not finish: the launch time is still measured)
            if i == 0 and k == 0: print("fetch:", e)
            res = None
        if i == 0 and k == 0 and os.environ.get("TIME_DUMP") and res is not None:
            np.savez(os.environ["TIME_DUMP"], contribs=res.contribs, chisq=res.chisq, moves=res.num_moves, iters=res.num_iter)
        if i >= 8:
            ms[k].append(pl.last_ms)
print("env DEBUG_FLAGS=%s UNCACHED=%s TICKS_PER_LAUNCH=%s LIB=%s: plan medians %s ms; moves mean %s" % (os.environ.get("MCSAS_DEBUG_FLAGS"), os.environ.get("MCSAS_HIP_UNCACHED"), os.environ.get("MCSAS_HIP_TICKS_PER_LAUNCH"), os.path.basename(os.environ.get("MCSAS_HIP_LIB", "default")),
      " ".join("%.4f" % np.median(x) for x in ms), plans[0].info))

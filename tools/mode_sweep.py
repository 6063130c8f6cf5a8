#!/usr/bin/env python3
"""MC steps/s of the three execution modes over the number of repetitions (sphere 512 q x 400 contribs):
the data behind the thresholds of MCSAS_EXEC_AUTO (mcsas_hip.hip, mode selection)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data

q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
for reps in (8, 16, 32, 64, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096):
    steps = max(1000, min(20000, 2000000 // reps))
    row = {"reps": reps, "steps": steps}
    for name, mode in (("wave", 1), ("workgroup", 2), ("pipeline", 3)):
        st = engine.Settings(n_contrib=400, n_reps=reps, max_iter=steps, conv_crit=0.0, max_retries=0, seed=1, exec_mode=mode)
        try:
            plan = engine.Plan(m.setup(), q, I, sig, st)
            plan.launch(); plan.fetch(want_arrays=False)
            plan.reseed(2); plan.launch(); plan.fetch(want_arrays=False)
            row[name] = round(plan.total_steps / (plan.last_ms * 1e-3) / 1e6, 1)
            plan.close()
        except Exception as e:
            row[name] = None
    print(json.dumps(row), flush=True)

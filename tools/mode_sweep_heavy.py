#!/usr/bin/env python3
"""MC steps/s of the three execution modes over the number of repetitions for a model whose rows cost an integral
(BASELINE config 3: cylinders 512 q x 400; config 5: Kholodenko 512 x 600): the data behind the heavy-row thresholds
of MCSAS_EXEC_AUTO.  The run part only: (two launches of `steps` - one launch of 0 steps) / 2."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcsas_amd import engine
from bench import workload

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = workload(cfg)
m, q, I, sig, n = wl["model"], wl["q"], wl["I"], wl["sigma"], wl["n"]
class FakeData:
    def __init__(self, q): self.q = q
for reps in (64, 128, 192, 256, 384, 512, 768, 1024, 2048):
    steps = 600
    row = {"reps": reps, "steps": steps}
    for name, mode in (("wave", 1), ("workgroup", 2), ("pipeline", 3)):
        t = {}
        try:
            for s in (0, steps):
                st = engine.Settings(n_contrib=n, n_reps=reps, max_iter=s, conv_crit=0.0, max_retries=0, seed=1, exec_mode=mode)
                plan = engine.Plan(m.setup(FakeData(q)) if hasattr(m, "setup") else m, q, I, sig, st)
                plan.launch(); plan.fetch(want_arrays=False)
                plan.reseed(2); plan.launch(); plan.fetch(want_arrays=False)
                t[s] = plan.last_ms; plan.close()
            row[name] = round(reps * steps / ((t[steps] - t[0]) * 1e-3) / 1e6, 2)
        except Exception as e:
            row[name] = None
    print(json.dumps(row), flush=True)

#!/usr/bin/env python3
"""Config 2's launch time per PLAN INSTANCE: k plans made one after the other in one process (each destroyed before the next is
made unless KEEP=1), n launches each — is the 3.41 / 3.52 ms split between runs a property of the process or of the plan's memory?
usage: tools/launch_variance.py [plans] [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
keep = os.environ.get("KEEP") == "1"
stream = 0
if os.environ.get("TORCH_STREAM") == "1":                    # (bench.py launches on a torch stream)
    import torch
    _ts = torch.cuda.Stream(); stream = _ts.cuda_stream
q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere()
m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
plans = []
for i in range(k):
    st = engine.Settings(n_contrib=400, n_reps=50, max_iter=20000, conv_crit=0.0, max_retries=0, seed=20250101 + i)
    plan = engine.Plan(m.setup(), q, I, sig, st)
    ms = []
    for j in range(n):
        plan.reseed(100 + j, 0); plan.launch(stream=stream); plan.fetch(want_arrays=False); ms.append(plan.last_ms)
    print("plan %d: launch ms min %.3f median %.3f max %.3f" % (i, min(ms[2:]), float(np.median(ms[2:])), max(ms[2:])), flush=True)
    if keep: plans.append(plan)
    else: plan.close()

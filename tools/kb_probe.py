#!/usr/bin/env python3
"""Window length of the row-queue pipeline (MCSAS_HIP_PIPE_KB, read per plan) on a BASELINE config: rate and whether the arrays equal the
default window's.  tools/kb_probe.py <config> <reps> <kb> [<kb> ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mcsas_amd import engine
cfg, reps = int(sys.argv[1]), int(sys.argv[2])
budget = {3: 10000, 4: 15000, 5: 10000}[cfg]
wl = bench.workload(cfg, 0)
ref = None
for kb in [0] + [int(x) for x in sys.argv[3:]]:
    if kb: os.environ["MCSAS_HIP_PIPE_KB"] = str(kb)
    else: os.environ.pop("MCSAS_HIP_PIPE_KB", None)
    st = engine.Settings(n_contrib=wl["n"], n_reps=reps, max_iter=budget, conv_crit=0.0, max_retries=0, seed=20250101)
    plan = engine.Plan(wl["model"].setup(), wl["q"], wl["I"], wl["sigma"], st)
    ms = []
    for i in range(5):
        plan.reseed(77, 0); plan.launch(); res = plan.fetch(); ms.append(plan.last_ms)
    if ref is None: ref = res
    same = all(np.array_equal(getattr(res, k), getattr(ref, k)) for k in ("contribs", "chisq", "num_moves", "fit"))
    print("config %d, %d reps, window %d (asked %d): %.2f ms  %.3e steps/s  arrays equal to the default window's: %s" %
          (cfg, reps, plan.info["window"], kb, np.median(ms[1:]), plan.total_steps / (np.median(ms[1:]) * 1e-3), same), flush=True)
    plan.close()

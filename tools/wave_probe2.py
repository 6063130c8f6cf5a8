#!/usr/bin/env python3
"""What the two rates of the wave-per-chain kernel depend on: footprint of the row cache, row length, cached rows or not."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import mcsas_amd
from mcsas_amd import engine

for nq, N, reps, steps, cache, nl in ((512, 400, 8192, 8000, 1, 8), (512, 400, 8192, 4000, 0, 8), (512, 100, 8192, 8000, 1, 8), (64, 400, 8192, 8000, 1, 8),
                                      (512, 400, 2048, 8000, 1, 8), (512, 400, 32768, 2000, 1, 8)):
    q, I, sig = bench.synthetic_data(nq)
    m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
    st = engine.Settings(n_contrib=N, n_reps=reps, max_iter=steps, conv_crit=0.0, max_retries=0, seed=1, exec_mode=engine.EXEC_WAVE, cache_intensities=cache)
    plan = engine.Plan(m.setup(), q, I, sig, st)
    time.sleep(0.5)
    ms = []
    with bench.ClockSampler(0) as cs:
        for i in range(nl):
            plan.reseed(100 + i, 0); plan.launch(); plan.fetch(want_arrays=False); ms.append(plan.last_ms)
    plan.close(); engine.release_cached_memory()
    ms = np.array(ms)
    c = cs.summary()
    print("nq %d N %d reps %d steps %d cache %d (rows %.1f GB): %s | fastest %.3e median %.3e | sclk %s power %s" % (
        nq, N, reps, steps, cache, reps * N * max(nq, 64) * 8 / 1e9, " ".join("%.1f" % x for x in ms), reps * steps / ms.min() * 1e3,
        reps * steps / np.median(ms) * 1e3, c.get("sclk_mhz", {}).get("median"), c.get("power_w", {}).get("median")), flush=True)

#!/usr/bin/env python3
"""A series of data sets (same shape), one analyse() after the other against engine.analyse_many (side by side on two streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
for nq, N, R, steps, crit in ((100, 300, 10, 100000, 1.0), (512, 400, 50, 20000, 0.0)):
    q, I, sig = synthetic_data(nq)
    m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
    probs = []
    for k in range(24):
        st = engine.Settings(n_contrib=N, n_reps=R, max_iter=steps if crit == 0.0 else 8000, conv_crit=0.0, max_retries=0, seed=100 + k)
        probs.append((m.setup(), q, I * (1 + 0.01 * k), sig, st))
    for rnd in range(2):
        t0 = time.perf_counter(); a = [engine.analyse(*p) for p in probs]; t1 = time.perf_counter()
        b = engine.analyse_many(probs); t2 = time.perf_counter()
        same = all(np.array_equal(x.contribs, y.contribs) for x, y in zip(a, b))
        print("%d q x %d x %d reps, 24 data sets: one after the other %.2f ms each, side by side %.2f ms each (identical: %s)" %
              (nq, N, R, (t1 - t0) / 24 * 1e3, (t2 - t1) / 24 * 1e3, same), flush=True)

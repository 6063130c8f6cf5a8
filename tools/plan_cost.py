#!/usr/bin/env python3
"""Host-side cost of one analysis: plan creation, launch + fetch, destruction (config 2 and the quick-start shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
for nq, N, R, steps in ((512, 400, 50, 20000), (100, 300, 10, 8000)):
    q, I, sig = synthetic_data(nq)
    m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
    st = engine.Settings(n_contrib=N, n_reps=R, max_iter=steps, conv_crit=0.0, max_retries=0, seed=7)
    setup = m.setup()
    for rnd in range(4):
        t0 = time.perf_counter(); pl = engine.Plan(setup, q, I, sig, st); t1 = time.perf_counter()
        pl.launch(); t2 = time.perf_counter(); res = pl.fetch(); t3 = time.perf_counter()
        ms = pl.last_ms
        del pl; t4 = time.perf_counter()
        print("%d q x %d x %d reps: create %.2f ms, launch (enqueue) %.2f, fetch (wait + copy) %.2f [kernels %.2f], destroy %.2f" %
              (nq, N, R, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, ms, (t4 - t3) * 1e3), flush=True)

#!/usr/bin/env python3
"""One analysis of configs 3 / 5 as ONE plan against the same repetitions split over two / three plans on the same device
(mcsas_problem.devices = (0, 0): concurrent host threads, plans and streams; results identical)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcsas_amd import engine
from bench import workload
for cfg, steps in ((5, 5000), (3, 5000), (4, 2500)):
    wl = workload(cfg)
    setup = wl["model"].setup()
    ref = None
    for devs in ((), (0, 0), (0, 0, 0), (), (0, 0)):
        st = engine.Settings(n_contrib=wl["n"], n_reps=wl["reps_gpu"], max_iter=steps, conv_crit=0.0, max_retries=0, seed=5,
                             exec_mode=engine.EXEC_PIPELINE, devices=devs)
        ts = []
        for i in range(3):
            t0 = time.perf_counter(); r = engine.analyse(setup, wl["q"], wl["I"], wl["sigma"], st); ts.append(time.perf_counter() - t0)
        if ref is None: ref = r
        same = np.array_equal(r.contribs, ref.contribs) and np.array_equal(r.num_moves, ref.num_moves)
        print("config %d, %d reps x %d steps, devices %s: %.2f ms wall (min of 3) -> %.3e steps/s, identical to one plan: %s" %
              (cfg, wl["reps_gpu"], steps, devs, min(ts) * 1e3, wl["reps_gpu"] * steps / min(ts), same), flush=True)

#!/usr/bin/env python3
"""More than 1024 q-points: one wavefront per chain (32 / 64 q slots per lane) against the q-split workgroup kernel
(chain_wide.h), sphere, 300 contributions.  usage: tools/wide_q_bench.py [steps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for nq in (2048, 4096, 8192, 16384):
    q, I, sig = synthetic_data(nq)
    m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
    for reps in (10, 256, 2048):
        row = {"nq": nq, "reps": reps}
        for name, mode in (("wave", engine.EXEC_WAVE), ("qsplit", engine.EXEC_WORKGROUP)):
            if mode == engine.EXEC_WAVE and nq > 4096:
                continue
            if reps * nq * 300 * 8 > 40e9:
                continue
            st = engine.Settings(n_contrib=300, n_reps=reps, max_iter=steps, conv_crit=0.0, max_retries=0, seed=3, exec_mode=mode)
            pl = engine.Plan(m.setup(), q, I, sig, st)
            best = 1e30
            for i in range(3):
                pl.reseed(10 + i, 0); pl.launch(); pl.fetch(want_arrays=False)
                best = min(best, pl.last_ms)
            row[name + "_ms"] = round(best, 3); row[name + "_steps_per_s"] = float("%.4g" % (reps * steps / (best * 1e-3)))
            del pl
        print(json.dumps(row), flush=True)

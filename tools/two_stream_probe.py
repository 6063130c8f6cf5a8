#!/usr/bin/env python3
"""Probe: G independent plans of 50/G chains each on G HIP streams (one host thread each) against one plan of 50
chains — do independent tick sequences overlap each other's launch gaps and tails?"""
import sys, os, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere()
m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
flags = int(os.environ.get("MCSAS_DEBUG_FLAGS", "0"))
def mk(reps, off):
    st = engine.Settings(n_contrib=400, n_reps=reps, max_iter=20000, conv_crit=0.0, max_retries=0, seed=2, exec_mode=0, debug_flags=flags)
    return engine.Plan(m.setup(), q, I, sig, st)
for G in (1, 2, 3, 5):
    R = 50 // G if G != 3 else 17
    plans = [mk(R, i * R) for i in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    def run(i, n):
        for _ in range(n):
            plans[i].launch(streams[i].cuda_stream); plans[i].fetch(want_arrays=False)
    for rnd in range(2):
        ths = [threading.Thread(target=run, args=(i, 6)) for i in range(G)]
        torch.cuda.synchronize(); t0 = time.time()
        for t in ths: t.start()
        for t in ths: t.join()
        torch.cuda.synchronize(); dt = time.time() - t0
    steps = G * R * 20000 * 6
    print("G=%d x %d chains: %.3f ms per launch round, %.3e steps/s" % (G, R, dt / 6 * 1e3, steps / dt), flush=True)

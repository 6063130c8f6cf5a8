#!/usr/bin/env python3
"""Config 2: K plans on K streams (their ticks overlap on the chip: one analysis' scan-bound early ticks beside another's
producer-bound late ones), optionally two result slots per plan, against one plan on one stream.
usage: tools/two_plan_overlap.py [analyses]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
st = engine.Settings(n_contrib=400, n_reps=50, max_iter=20000, conv_crit=0.0, max_retries=0, seed=20250101)
def run(K, slots):
    plans = [engine.Plan(m.setup(), q, I, sig, st) for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    lanes = [(k, s) for s in range(slots) for k in range(K)]          # (plan, slot) in launch order
    ms = []
    def go(count):
        pend = []
        for i in range(count):
            k, s = lanes[i % len(lanes)]
            while (k, s) in pend or len(pend) >= len(lanes):
                kk, ss = pend.pop(0); plans[kk].fetch(slot=ss); ms.append(plans[kk].last_ms)
            plans[k].reseed(1000 + i, 0); plans[k].launch(stream=streams[k].cuda_stream, slot=s); pend.append((k, s))
        while pend:
            kk, ss = pend.pop(0); plans[kk].fetch(slot=ss); ms.append(plans[kk].last_ms)
    go(2 * len(lanes) + 2); torch.cuda.synchronize(); del ms[:]
    t0 = time.perf_counter(); go(n); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d plan(s) on %d stream(s), %d slot(s) each: %.4e steps/s   (kernel time per analysis median %.3f ms)" % (K, K, slots, n * 1e6 / dt, np.median(ms)), flush=True)
    del plans
for K, slots in ((1, 2), (2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (1, 2), (2, 2)):
    run(K, slots)

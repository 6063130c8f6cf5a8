#!/usr/bin/env python3
"""Whole-job rate of config 2 with two plans x two result slots on two streams (bench.py's scheme), for each library given:
the producers' time is what limits that regime, so this — not the single-stream tick sum — is where a producer-side change shows.
usage: tools/overlap_ab.py libA.so [libB.so ...] [analyses]   (each library in a process of its own, twice, interleaved)"""
import os, subprocess, sys
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
n = [a for a in sys.argv[1:] if not a.endswith(".so")]
n = n[0] if n else "160"
child = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data
n = int(sys.argv[1])
q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
st = engine.Settings(n_contrib=400, n_reps=50, max_iter=20000, conv_crit=0.0, max_retries=0, seed=20250101)
plans = [engine.Plan(m.setup(), q, I, sig, st) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
lanes = [(k, s) for s in range(2) for k in range(2)]
def go(count):
    pend = []
    for i in range(count):
        k, s = lanes[i % 4]
        while (k, s) in pend or len(pend) >= 4:
            kk, ss = pend.pop(0); plans[kk].fetch(slot=ss, want_arrays=False)
        plans[k].reseed(1000 + i, 0); plans[k].launch(stream=streams[k].cuda_stream, slot=s); pend.append((k, s))
    while pend:
        kk, ss = pend.pop(0); plans[kk].fetch(slot=ss, want_arrays=False)
go(12); torch.cuda.synchronize()
t0 = time.perf_counter(); go(n); torch.cuda.synchronize()
print("%.4e" % (n * 1e6 / (time.perf_counter() - t0)))
'''
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, MCSAS_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", child, n], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)

#!/usr/bin/env python3
"""One wavefront per chain with thousands of chains: per-launch times in launch order with sclk / power beside them, for the
eight-chains-per-workgroup variant and (tuning library, bit 21) single-wave workgroups, at several chain counts and step budgets.
    python3 tools/wave_probe.py > gpurun_out/wave_probe.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mcsas_amd import engine

wl = bench.workload(2)
setup = wl["model"].setup()
CASES = ((8192, 2000, 0, 24), (8192, 2000, 1 << 21, 24), (4096, 2000, 0, 16), (16384, 1000, 0, 16), (8192, 500, 0, 40), (8192, 8000, 0, 8))
if len(sys.argv) > 1 and sys.argv[1] == "long":
    CASES = ((8192, 8000, 0, 8), (8192, 8000, 1 << 21, 8), (8192, 20000, 0, 5), (8192, 20000, 1 << 21, 5), (2048, 20000, 0, 5), (2048, 20000, 1 << 21, 5))
if len(sys.argv) > 1 and sys.argv[1] == "pad":
    CASES = [(8192, 8000, 0, 10, pad) for pad in (0, 1, 3, 0, 1, 7)]
for case in CASES:
    reps, steps, flags, nl = case[:4]
    if len(case) > 4:
        os.environ["MCSAS_HIP_CACHE_PAD_ROWS"] = str(case[4]); print("pad rows", case[4])
    st = engine.Settings(n_contrib=400, n_reps=reps, max_iter=steps, conv_crit=0.0, max_retries=0, seed=1, exec_mode=engine.EXEC_WAVE, debug_flags=flags)
    plan = engine.Plan(setup, wl["q"], wl["I"], wl["sigma"], st)
    time.sleep(1.0)                                        # (let the chip cool between cases)
    rows = []
    with bench.ClockSampler(0) as cs:
        t0 = time.perf_counter()
        for i in range(nl):
            plan.reseed(100 + i, 0); plan.launch(); plan.fetch(want_arrays=False)
            rows.append((time.perf_counter() - t0, plan.last_ms, plan.total_steps))
    plan.close(); engine.release_cached_memory(tuning=bool(flags))
    ms = np.array([r[1] for r in rows])
    print("reps %d steps %d flags %d: %s" % (reps, steps, flags, " ".join("%.1f" % m for m in ms)))
    print("   rate fastest %.3e median %.3e; init-only estimate n/a; clocks %s" % (reps * steps / ms.min() * 1e3, reps * steps / np.median(ms) * 1e3, cs.summary()))
    # sclk / power trace: one sample per 50 ms
    tr = [(s[0][0], s[0][1]) for s in cs.samples]
    print("   sclk MHz:", " ".join("%d" % (f / 1e6) for f, p in tr if f))
    print("   power W:", " ".join("%d" % (p / 1e6) for f, p in tr if p))

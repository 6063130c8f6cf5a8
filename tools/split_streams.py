#!/usr/bin/env python3
"""One analysis of a heavy config as ONE plan vs split over k concurrent sub-plans on the same device (mcsas_problem.devices with
the device listed k times: every block of chains runs from its own host thread, plan and stream) — the tick kernels of the blocks
fill each other's tails.  usage: tools/split_streams.py [config] [mc_steps]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from mcsas_amd import engine
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
wl = bench.workload(cfg, 0)
setup = wl["model"].setup()
for k in (1, 2, 3, 4):
    st = engine.Settings(n_contrib=wl["n"], n_reps=wl["reps_gpu"], max_iter=steps, conv_crit=0.0, max_retries=0, seed=5,
                         devices=(0,) * k if k > 1 else ())
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = engine.analyse(setup, wl["q"], wl["I"], wl["sigma"], st)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    print("config %d, %d chains x %d steps, %d sub-plan(s): %.2f ms wall, %.4g steps/s (analyse() incl. plan set-up and fetch)" % (
        cfg, wl["reps_gpu"], steps, k, t * 1e3, wl["reps_gpu"] * steps / t))

#!/usr/bin/env python3
"""Paired A/B of two builds of libmcsas_hip.so on config 2 in ONE process.  K plans per build; the launches of all 2K
plans alternate (A0 B0 A1 B1 ... A0 ...), so that the chip's clock / power drift (launch times wander by +-4 % over seconds)
hits both builds alike.  Several plans per build because a plan's speed also depends on where its buffers landed
(tools/placement_probe.py: one plan in ten is 2-4 % slower than its siblings for as long as it lives): compare the
builds' BEST plans and their medians over plans.
usage: tools/ab_pair.py libA.so libB.so [launches per plan] [plans per build]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from mcsas_amd import engine, _lib
from bench import synthetic_data

paths = [os.path.abspath(p) for p in sys.argv[1:3]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
q, I, sig = synthetic_data(512)
m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
st = engine.Settings(n_contrib=400, n_reps=50, max_iter=int(os.environ.get("AB_STEPS", "20000")), conv_crit=0.0, max_retries=0, seed=20250101)
plans = []
for k in range(K):
    for which, p in enumerate(paths):
        _lib._libs.pop(False, None)
        _lib.LIB_PATH = p
        os.environ["MCSAS_HIP_LIB"] = p
        plans.append((which, engine.Plan(m.setup(), q, I, sig, st)))
ms = [[] for _ in plans]
for i in range(n + 10):
    for k, (_, pl) in enumerate(plans):
        pl.reseed(1000 + i, 0); pl.launch(); pl.fetch(want_arrays=False)
        if i >= 10:
            ms[k].append(pl.last_ms)
med = [[], []]
for k, (which, _) in enumerate(plans):
    med[which].append(float(np.median(ms[k])))
for which in (0, 1):
    print("%s %s: plan medians %s ms" % ("AB"[which], os.path.basename(paths[which]), " ".join("%.4f" % v for v in med[which])))
print("B/A - 1: best plans %+.2f %%, median plans %+.2f %%" % (100 * (min(med[1]) / min(med[0]) - 1), 100 * (np.median(med[1]) / np.median(med[0]) - 1)))

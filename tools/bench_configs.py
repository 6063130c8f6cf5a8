#!/usr/bin/env python3
"""Throughput of the other BASELINE configurations and of the many-chain (wavefront) mode, for
DESIGN.md.  Not the contract bench (that is bench.py, config 2); same measurement style:
fixed step budget, convergenceCriterion = 0, HIP-event time of the launch."""
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data

CASES = [
    # name, model class, active params, lo, hi, nq, n_contrib, reps, mc_steps, mode
    ("cfg2 sphere 512x400, 50 reps (pipeline)", mcsas_amd.Sphere, ["radius"], None, None, 512, 400, 50, 20000, 0),
    ("cfg2 sphere 512x400, 50 reps (workgroup)", mcsas_amd.Sphere, ["radius"], None, None, 512, 400, 50, 20000, 2),
    ("sphere 512x400, 8192 reps (wavefront, throughput mode)", mcsas_amd.Sphere, ["radius"], None, None, 512, 400, 8192, 2000, 1),
    ("sphere 512x400, 1024 reps (workgroup)", mcsas_amd.Sphere, ["radius"], None, None, 512, 400, 1024, 4000, 2),
    ("cfg3 cylinders 512x400, 25 reps (pipeline)", mcsas_amd.CylindersIsotropic, ["radius", "aspect"], [1e-9, 0.5], [1e-7, 20.0], 512, 400, 25, 1000, 0),
    ("cfg4 core-shell ellipsoid 1024x1000, 50 reps (pipeline)", mcsas_amd.EllipsoidalCoreShell, ["a", "b", "t"], [1e-9, 2e-9, 2e-10], [1e-7, 2e-7, 1e-8], 1024, 1000, 50, 600, 0),
    ("cfg5 Kholodenko 512x600, 13 reps (pipeline)", mcsas_amd.Kholodenko, ["radius", "lenKuhn", "lenContour"], None, None, 512, 600, 13, 200, 0),
]


def main():
    out = []
    for name, cls, active, lo, hi, nq, n, reps, steps, mode in CASES:
        q, I, sig = synthetic_data(nq)
        m = cls()
        for p in m.params():
            if hasattr(p, "setActive"):
                p.setActive(p.name() in active)
        if lo is None and cls is mcsas_amd.Sphere:
            lo, hi = [np.pi / q.max()], [np.pi / q.min()]
        if lo is not None:
            for nme, l, h in zip(active, lo, hi):
                getattr(m, nme).setActiveRange((l, h))
        st = engine.Settings(n_contrib=n, n_reps=reps, max_iter=steps, conv_crit=0.0, max_retries=0, seed=1, exec_mode=mode)
        plan = engine.Plan(m.setup(), q, I, sig, st)
        plan.launch(); plan.fetch(want_arrays=False)          # warm-up
        plan.reseed(2)
        plan.launch(); res = plan.fetch()
        ms, total = plan.last_ms, plan.total_steps
        rec = dict(case=name, mc_steps_per_s=total / (ms * 1e-3), ms=ms, mc_steps=total, chisq_median=float(np.median(res.chisq)), **plan.info)
        out.append(rec)
        print(json.dumps(rec), flush=True)
        plan.close()
    return out


if __name__ == "__main__":
    main()

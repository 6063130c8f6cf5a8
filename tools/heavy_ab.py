import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
libs = sys.argv[1:]
import bench
from mcsas_amd import engine, _lib
out = {}
for cfg, budget in ((3, 10000), (4, 15000), (5, 10000)):
    wl = bench.workload(cfg, 0)
    for lib in libs:
        _lib._libs.pop(False, None); _lib.LIB_PATH = os.path.abspath(lib); os.environ["MCSAS_HIP_LIB"] = os.path.abspath(lib)
        st = engine.Settings(n_contrib=wl["n"], n_reps=wl["reps_gpu"], max_iter=budget, conv_crit=0.0, max_retries=0, seed=20250101)
        plan = engine.Plan(wl["model"].setup(), wl["q"], wl["I"], wl["sigma"], st)
        ms = []
        for i in range(6):
            plan.reseed(77 + i, 0); plan.launch(); res = plan.fetch(); ms.append(plan.last_ms)
        print(cfg, os.path.basename(lib), "ms", " ".join("%.2f" % m for m in ms), "steps/s %.3e" % (plan.total_steps / (np.median(ms[1:]) * 1e-3)), "chisq med %.6g" % np.median(res.chisq), flush=True)
        plan.close()

#!/usr/bin/env python3
"""MC steps/s of the three execution modes over (q points) x (contributions) x (repetitions), sphere model: the table
behind MCSAS_EXEC_AUTO (mcsas_hip.hip, mode selection).  Fixed budget, convergenceCriterion 0, HIP-event time of the second
launch of a plan (chain initialisation included, as a user pays it).  One JSON line per point on stdout; `auto` = what
MCSAS_EXEC_AUTO picked and delivered.  The reference's default shape (mcsasparameters.json: 300 contributions, 10
repetitions; 100 q after the default rebinning) is in the grid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from mcsas_amd import engine
from bench import synthetic_data

QS = [int(x) for x in os.environ.get("SWEEP_Q", "100,256,512,1024").split(",")]
NS = [int(x) for x in os.environ.get("SWEEP_N", "200,300,1000").split(",")]
RS = [int(x) for x in os.environ.get("SWEEP_R", "1,10,50,200,1000").split(",")]
for nq in QS:
    q, I, sig = synthetic_data(nq)
    m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
    for n in NS:
        for reps in RS:
            steps = int(max(2000, min(20000, 4000000 // reps)))
            row = {"nq": nq, "n_contrib": n, "reps": reps, "steps": steps}
            for name, mode in (("wave", 1), ("workgroup", 2), ("pipeline", 3), ("auto", 0)):
                st = engine.Settings(n_contrib=n, n_reps=reps, max_iter=steps, conv_crit=0.0, max_retries=0, seed=1, exec_mode=mode)
                try:
                    plan = engine.Plan(m.setup(), q, I, sig, st)
                    plan.launch(); plan.fetch(want_arrays=False)
                    ms = []
                    for k in range(2):
                        plan.reseed(2 + k); plan.launch(); plan.fetch(want_arrays=False); ms.append(plan.last_ms)
                    row[name] = round(plan.total_steps / (min(ms) * 1e-3) / 1e6, 2)
                    if mode == 0:
                        row["auto_mode"] = plan.info["exec_mode"]
                    plan.close()
                except Exception as e:
                    row[name] = None
            best = max((row[k] or 0.0, k) for k in ("wave", "workgroup", "pipeline"))
            row["best"] = best[1]
            row["auto_vs_best"] = round((row["auto"] or 0.0) / best[0], 3) if best[0] else None
            print(json.dumps(row), flush=True)

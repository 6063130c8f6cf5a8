#!/usr/bin/env python3
"""Wall time of BASELINE configs 3-5 run to their criterion (bench.convergence_run) — chains that converge at different times
leave their producer workgroups to the others (MCSAS_HIP_PIPE_HELP=0 switches that off).  usage: tools/conv_heavy.py [configs]"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
from mcsas_amd import engine
for cfg in [int(x) for x in (sys.argv[1:] or ["3", "4", "5"])]:
    wl = bench.workload(cfg, 0)
    setup = wl["model"].setup()
    for crit_scale in (1.0,):
        r = bench.convergence_run(engine, setup, wl["q"], wl["I"], wl["sigma"], wl["n"], wl["reps_gpu"], 0, 0, wl["chisq_of_truth"])
        a = r.get("at_reachable_criterion") or {}
        print("config %d: criterion %.3g wall %.4f s converged %d/%d steps mean %.0f | reachable crit %s wall %s converged %s steps %s" % (
            cfg, r["criterion"], r["wall_s"], r["converged"], r["reps"], r["steps_mean"], a.get("criterion"), a.get("wall_s"), a.get("converged"), a.get("steps_mean")))

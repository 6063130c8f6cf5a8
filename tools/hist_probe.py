#!/usr/bin/env python3
"""Where McSAS.histogram() of config 2 spends its time: the library call (mcsas_hip_histogram_prep: uploads, three kernels,
downloads) and the host half (fractions, bins, CDF, moments).  python3 tools/hist_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, mcsas_amd
from mcsas_amd import engine
wl = bench.workload(2)
q = wl["q"]
m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
lo, hi = m.radius.activeRange()
m.radius.histograms().append(mcsas_amd.Histogram(m.radius, lo, hi, binCount=50, xscale='log', yweight='vol'))
algo = mcsas_amd.McSAS(seed=1)
algo.numContribs.setValue(400); algo.numReps.setValue(50); algo.maxIterations.setValue(20000); algo.convergenceCriterion.setValue(0.0)
algo.showIncomplete.setValue(True); algo.maxRetries = mcsas_amd.mcsas._Setting("maxRetries", 0)
algo.model = m; algo.data = mcsas_amd.SASData(q, wl["I"], wl["sigma"])
import logging; logging.disable(logging.WARNING)
algo.result = []; algo.stop = False
algo.analyse()
contribs = algo.result[0]["contribs"]
setup = mcsas_amd.scatteringmodels.setup_from_model(m, algo.data)
sig = np.array(algo.data.f.binnedDataU, dtype=float)
for trial in range(6):
    t0 = time.perf_counter()
    out = engine.histogram_prep(setup, q, algo.data.f.binnedData, sig, contribs, algo.compensationExponent())
    t1 = time.perf_counter()
    algo.histogram()
    t2 = time.perf_counter()
    algo.result = []; algo.analyse(); t3 = time.perf_counter()
    print("histogram_prep %.3f ms | histogram() %.3f ms | analyse() %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)

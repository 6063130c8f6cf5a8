#!/usr/bin/env python3
"""Wall time of McSAS.calc() end to end (analyse + histogram) on BASELINE config 2's shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcsas_amd
from bench import synthetic_data

q, I, sig = synthetic_data(512)
d = mcsas_amd.SASData(q, I, sig)
m = mcsas_amd.Sphere(); m.radius.setActiveRange((np.pi / q.max(), np.pi / q.min()))
m.radius.histograms().append(mcsas_amd.Histogram(m.radius, np.pi / q.max(), np.pi / q.min(), binCount=50, xscale='log', yweight='vol'))
algo = mcsas_amd.McSAS(seed=1)
algo.numContribs.setValue(400); algo.numReps.setValue(50); algo.convergenceCriterion.setValue(2.0)
algo.model, algo.data = m, d
for it in range(3):
    algo.result = []
    t0 = time.perf_counter(); algo.analyse(); t1 = time.perf_counter(); algo.histogram(); t2 = time.perf_counter()
    print("run %d: analyse %.1f ms, histogram %.1f ms, chisq max %.3f" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, algo.details.chisq.max()))

# where histogram() spends its time
from mcsas_amd import engine
from mcsas_amd.scatteringmodels import setup_from_model
contribs = algo.result[0]['contribs']
setup = setup_from_model(m, d)
for it in range(2):
    t0 = time.perf_counter()
    out = engine.histogram_prep(setup, d.q, d.f.binnedData, d.f.binnedDataU, contribs, 0.6666666)
    t1 = time.perf_counter()
    for paramIndex, param in enumerate(m.activeParams()):
        param.histograms().calc(contribs, paramIndex, algo.fractions)
    t2 = time.perf_counter()
    print("histogram_prep %.1f ms, Histogram.calc (host binning) %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))

# where analyse() spends its time
st = engine.Settings(n_contrib=400, n_reps=50, max_iter=100000, conv_crit=2.0, max_retries=5, seed=1)
for it in range(3):
    t0 = time.perf_counter()
    plan = engine.Plan(setup, d.q, d.f.binnedData, d.f.binnedDataU, st)
    t1 = time.perf_counter()
    plan.launch(); res = plan.fetch()
    t2 = time.perf_counter()
    plan.close()
    t3 = time.perf_counter()
    print("plan create %.2f ms, launch+fetch %.2f ms (device %.2f ms), destroy %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, plan.last_ms if False else 0.0, (t3 - t2) * 1e3))

"""Series of data sets through one algorithm object: the computational core of the reference's
Calculator (gui/calc.py:271-379) without its log files, HDF5 archive and plots: for every data set
`algo.data = dataset; algo.calc()`, then the moments of every histogram are collected per
(sample, parameter, range, weighting) the way `_updateSeries` does (:331-349)."""
from __future__ import annotations

from collections import OrderedDict


def run_series(algo, datasets, keys=None, replays=None, overlap=False):
    """-> (results, series).  `results[i]` is `algo.result[0]` of data set i (None when nothing
    converged); `series[(param, lower, upper, yweight)]` is the list of `(key_i, moments.fields)`.
    `keys`: the series key value of each data set (default: its index).  `replays`: per data set, the uniform
    streams of its repetitions (tests replaying the reference: tests/golden/g15_series.npz).  `overlap`: the data sets' analyses
    run side by side on the device instead of one after the other (same results; algo.seed should be set: an unseeded algo draws a
    new seed per data set either way)."""
    if algo.model is None:
        raise ValueError("no model set")
    results, series = [], OrderedDict()
    chains = None
    if overlap and algo.model.paramCount():
        algo.stop = False                                    # calc() resets it per data set (mcsas.py:152); here once, for the batch
        # the Monte-Carlo part of every data set first, side by side on the device (engine.analyse_many: the data sets' analyses are
        # independent); the loop below then only stores each result and takes its histograms — the same numbers as one after the other
        from . import engine
        problems = []
        for i, data in enumerate(datasets):
            algo.data = data
            problems.append(algo._problem(replay=None if replays is None else replays[i]))
        chains = engine.analyse_many(problems)
    for i, data in enumerate(datasets):
        key = i if keys is None else keys[i]
        algo.data = data
        if chains is None:
            algo.calc(replay=None if replays is None else replays[i])
        else:
            # (McSAS.stop is left as the analyses saw it: _store tells "stop pressed" from "criterion not reached" by it)
            algo.result = []
            algo._store(chains[i])
            if len(algo.result):
                algo.histogram()
        if not (isinstance(algo.result, list) and len(algo.result)):
            results.append(None)
            continue
        results.append(algo.result[0])
        for p in algo.model.activeParams():
            for h in p.histograms():
                uid = (p.name(),) + tuple(h.xrange) + (h.yweight,)
                series.setdefault(uid, []).append((key, h.moments.fields))
        if algo.stop:
            break
    return results, series

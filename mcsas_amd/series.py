"""Series of data sets through one algorithm object: the computational core of the reference's
Calculator (gui/calc.py:271-379) without its log files, HDF5 archive and plots: for every data set
`algo.data = dataset; algo.calc()`, then the moments of every histogram are collected per
(sample, parameter, range, weighting) the way `_updateSeries` does (:331-349)."""
from __future__ import annotations

from collections import OrderedDict


def run_series(algo, datasets, keys=None, replays=None):
    """-> (results, series).  `results[i]` is `algo.result[0]` of data set i (None when nothing
    converged); `series[(param, lower, upper, yweight)]` is the list of `(key_i, moments.fields)`.
    `keys`: the series key value of each data set (default: its index).  `replays`: per data set, the uniform
    streams of its repetitions (tests replaying the reference: tests/golden/g15_series.npz)."""
    if algo.model is None:
        raise ValueError("no model set")
    results, series = [], OrderedDict()
    for i, data in enumerate(datasets):
        key = i if keys is None else keys[i]
        algo.data = data
        algo.calc(replay=None if replays is None else replays[i])
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

"""Parameter / FitParameter / NumberGenerator / Histogram: the slice of the reference's parameter
system the hot path and its callers touch, with the reference's method names.

  reference: bases/algorithm/parameter.py:205-330,390-525 (ParameterBase/Numerical/Float),
             utils/parameter.py:577-743 (FitParameter*), bases/algorithm/numbergenerator.py,
             utils/parameter.py:187-538 (Histogram) and :20-184 (Moments, VectorResult).
Values are SI throughout (the reference converts display units on entry; units.py is out of scope).
Histogram binning is host-side post-processing; the per-contribution intensities
it needs come from the GPU (engine.model_calc / engine.observability).
"""
from __future__ import annotations

import numpy as np


# ------------------------------------------------------------------ number generators
class NumberGenerator(object):
    """numbergenerator.py:14-22.  `kind` is the MCSAS_GEN_* code the kernels implement."""
    kind = 0

    @classmethod
    def get(cls, count=1):
        raise NotImplementedError


class RandomUniform(NumberGenerator):
    kind = 0

    @classmethod
    def get(cls, count=1):                                   # numbergenerator.py:28-31
        return np.random.uniform(size=count)


class RandomExponential(NumberGenerator):
    kind = 1
    lower, upper = 0., 1.

    @classmethod
    def get(cls, count=1):                                   # numbergenerator.py:168-175
        rs = 10**(np.random.uniform(cls.lower, cls.upper, count))
        return (rs - 1) / (10**(cls.upper - cls.lower))


class RandomExponential1(RandomExponential):
    pass


class RandomExponential2(RandomExponential):
    kind = 2
    upper = 2.


class RandomExponential3(RandomExponential):
    kind = 3
    upper = 3.


def generator_kind(gen):
    """MCSAS_GEN_* of a generator class: ours by attribute, the reference's by class name."""
    k = getattr(gen, "kind", None)
    if k is not None:
        return int(k)
    name = getattr(gen, "__name__", str(gen))
    table = {"RandomUniform": 0, "RandomExponential": 1, "RandomExponential1": 1,
             "RandomExponential2": 2, "RandomExponential3": 3}
    if name not in table:
        raise ValueError("generator %s has no device implementation" % name)
    return table[name]


# ------------------------------------------------------------------ parameters
class Parameter(object):
    """A named model value with a valueRange; callable like the reference's (`p()` -> value)."""

    def __init__(self, name, value, displayName=None, valueRange=None, **_ignored):
        self._name = name
        self._valueRange = (-np.inf, np.inf)                  # a parameter declared without a range has none
        self._displayName = displayName or name
        self._value = value
        if valueRange is not None:
            self.setValueRange(valueRange)                    # (clips the constructor's value into the range like the reference)

    def name(self):
        return self._name

    def displayName(self):
        return self._displayName

    def value(self):
        return self._value

    __call__ = value

    def valueRange(self):
        return self._valueRange

    def setValueRange(self, newRange):
        # bases/algorithm/parameter.py:420-433: infinities are stored as +-1e200 ("as good as inf")
        self._valueRange = (max(min(newRange), -1e200), min(max(newRange), 1e200))
        # ... and the current value is brought into the new range (:431-433: setValue(clip()))
        v = getattr(self, "_value", None)
        if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, (bool, np.bool_)):
            self._value = type(v)(np.clip(v, self._valueRange[0], self._valueRange[1])) if isinstance(v, (int, np.integer)) else float(np.clip(v, self._valueRange[0], self._valueRange[1]))

    def min(self):
        return self._valueRange[0]

    def max(self):
        return self._valueRange[1]

    def clip(self, value=None):                              # parameter.py:489-495
        if value is None:
            value = self._value
        return np.clip(value, self.min(), self.max())

    def setValue(self, newValue, clip=True):                 # parameter.py:405-414
        if newValue is None:
            return
        if isinstance(newValue, (bool, np.bool_)):
            self._value = bool(newValue)
            return
        self._value = float(self.clip(newValue)) if clip else float(newValue)

    def isActive(self):
        return False


class Histograms(list):
    def calc(self, *args):                                   # utils/parameter.py:562-565
        for h in self:
            h.calc(*args)


class FitParameter(Parameter):
    """utils/parameter.py:577-743: adds isActive / activeRange / generator / histograms / activeValues."""

    def __init__(self, name, value, displayName=None, valueRange=None, activeRange=None,
                 generator=RandomUniform, **kw):
        super().__init__(name, value, displayName, valueRange, **kw)
        self._isActive = False
        self._activeRange = None
        self._generator = generator
        self._histograms = Histograms()
        self._activeValues = []
        if activeRange is not None:
            self.setActiveRange(activeRange)

    def isActive(self):
        return self._isActive

    def setIsActive(self, isActive):
        self._isActive = bool(isActive)

    setActive = setIsActive

    def setActiveRange(self, newRange):                      # utils/parameter.py:615-624
        if len(newRange) != 2:
            raise ValueError("Active ranges have to consist of two values!")
        r = self.clip(np.asarray(newRange, dtype=float))
        self._activeRange = (float(min(r)), float(max(r)))
        for h in self._histograms:
            h.updateRange()

    def activeRange(self):
        return self._activeRange if self._activeRange is not None else self.valueRange()

    def generator(self):
        return self._generator

    def setGenerator(self, gen):
        self._generator = gen if isinstance(gen, type) else RandomUniform

    def histograms(self):
        return self._histograms

    def activeValues(self):
        return self._activeValues

    def activeVal(self, index=None):
        if index is None:
            return self._activeValues
        return self._activeValues[index % len(self._activeValues)]

    def setActiveVal(self, val, index=None):                 # utils/parameter.py:666-694
        if not self.isActive():
            return
        vals = self._activeValues
        if index is None:
            index = len(vals)
        elif index < 0:
            index = len(vals) + index + 1
        while len(vals) <= index:
            vals.append(None)
        vals[index] = val

    def generate(self, lower=None, upper=None, count=1):     # utils/parameter.py:715-728, parameter.py:66-84
        lo = min(self.activeRange()) if lower is None else lower
        hi = max(self.activeRange()) if upper is None else upper
        lo, hi = max(self.min(), lo), min(self.max(), hi)
        return self._generator.get(count) * (hi - lo) + lo


def isActiveFitParam(param):
    """utils/parameter.py:573-576; duck-typed so the reference's own parameter objects work too."""
    f = getattr(param, "isActive", None)
    return bool(f()) if callable(f) else False


# ------------------------------------------------------------------ histogram (post-fit)
class VectorResult(object):
    """utils/parameter.py:156-184."""

    def __init__(self, vec):
        assert vec.ndim == 2
        self.full = vec
        self.mean = vec.mean(axis=1)
        self.std = vec.std(axis=1, ddof=1 if len(vec) > 1 else 0)


class Moments(object):
    """utils/parameter.py:20-122 (partial intensities :124-154 are not on the path)."""
    @staticmethod
    def fieldNames():
        return ("totalValue", "totalValueStd", "mean", "meanStd", "variance", "varianceStd",
                "skew", "skewStd", "kurtosis", "kurtosisStd")

    def __init__(self, contribs, paramIndex, valueRange, fraction):
        numReps = contribs.shape[2]
        # (repetition-major copies: the ordered sums below then run along contiguous rows)
        vals = np.ascontiguousarray(contribs[:, paramIndex, :].T)
        fraction = np.ascontiguousarray(np.asarray(fraction).T)
        lo, hi = min(valueRange), max(valueRange)
        # All repetitions at once.  The reference sums the valid contributions of a repetition with the builtin sum(), i.e. in
        # contribution order; a cumulative sum down the contribution axis with the invalid entries replaced by +0.0 adds the
        # same numbers in the same order (x + 0.0 == x bit for bit).  The powers are products (d*d*d, (d*d)*(d*d)): numpy's
        # `**3` / `**4` go through pow(), whose last bit differs between its scalar and its SIMD array path on one machine already
        # (and 40 000 calls of it were most of histogram()'s time); the moments are compared at 1e-6.
        valid = (vals > lo) & (vals < hi)
        anyv = valid.any(axis=1)

        def ssum(x):
            return np.cumsum(np.where(valid, x, 0.0), axis=1)[:, -1] if x.shape[1] else np.zeros(numReps)

        with np.errstate(invalid='ignore', divide='ignore', over='ignore'):
            tot = ssum(fraction)
            mu = ssum(vals * fraction)
            mu = np.where(tot != 0, mu / np.where(tot != 0, tot, 1.0), mu)
            dev = vals - mu[:, None]
            dev2 = dev * dev
            var = ssum(dev2 * fraction) / tot
            sigma = np.sqrt(np.abs(var))
            # (utils/parameter.py:106-107 skips only an exact zero: a NaN product — all-zero weighting, e.g. 'surf' on a model
            # without surface() — goes on and gives NaN skew / kurtosis, like the variance)
            ok = anyv & ((tot * sigma) != 0.0)
            sigma2 = sigma * sigma
            skw = np.where(ok, ssum(dev2 * dev * fraction) / (tot * (sigma2 * sigma)), 0.0)
            krt = np.where(ok, ssum(dev2 * dev2 * fraction) / (tot * (sigma2 * sigma2)), 0.0)
        val = np.where(anyv, tot, 0.0); mu = np.where(anyv, mu, 0.0); var = np.where(anyv, var, 0.0)
        ddof = 1 if numReps > 1 else 0
        self.total = (val.mean(), val.std(ddof=ddof)); self.mean = (mu.mean(), mu.std(ddof=ddof))
        self.variance = (var.mean(), var.std(ddof=ddof)); self.skew = (skw.mean(), skw.std(ddof=ddof))
        self.kurtosis = (krt.mean(), krt.std(ddof=ddof))

    @classmethod
    def fromRepetitions(cls, val, mu, var, skw, krt):
        """The per-repetition moments worked out elsewhere (mcsas_hip_histogram) -> their mean / std over the repetitions."""
        self = cls.__new__(cls)
        ddof = 1 if len(val) > 1 else 0
        self.total = (val.mean(), val.std(ddof=ddof)); self.mean = (mu.mean(), mu.std(ddof=ddof))
        self.variance = (var.mean(), var.std(ddof=ddof)); self.skew = (skw.mean(), skw.std(ddof=ddof))
        self.kurtosis = (krt.mean(), krt.std(ddof=ddof))
        return self

    @property
    def fields(self):
        return self.total + self.mean + self.variance + self.skew + self.kurtosis


class Histogram(object):
    """utils/parameter.py:187-538: bins, CDF, observability and moments of one parameter."""

    def __init__(self, param, lower, upper, binCount=50, xscale=None, yweight=None, autoFollow=True):
        self.param = param
        self.binCount = max(0, int(binCount))
        self.xrange = (float(lower), float(upper))
        self.xscale = xscale
        self.yweight = yweight
        self.autoFollow = bool(autoFollow)
        self.xLowerEdge = self.xMean = self.xWidth = None
        self.bins = self.cdf = self.observability = self.moments = None

    @staticmethod
    def xscaling():
        return ('lin', 'log')

    @staticmethod
    def yweighting():
        return ('vol', 'num', 'int', 'surf')

    @property
    def xscale(self):
        return self._xscale

    @xscale.setter
    def xscale(self, kind):                                   # :244-249 (invalid -> 'log')
        kind = str(kind).strip()
        self._xscale = kind if kind in self.xscaling() else 'log'

    @property
    def yweight(self):
        return self._yweight

    @yweight.setter
    def yweight(self, kind):                                  # :255-259 (invalid -> 'vol')
        kind = str(kind).strip()
        self._yweight = kind if kind in self.yweighting() else 'vol'

    @property
    def xrange(self):
        return self._xrange

    @xrange.setter
    def xrange(self, valueRange):                             # :265-270
        lo, hi = min(valueRange), max(valueRange)
        lo, hi = max(self.param.min(), lo), min(self.param.max(), hi)
        self._xrange = (lo, hi)

    @property
    def lower(self):
        return self._xrange[0]

    @property
    def upper(self):
        return self._xrange[1]

    def updateRange(self):                                    # :292-297
        if self.autoFollow:
            self.xrange = self.param.activeRange()
        self.xrange = self.xrange

    def _setXLowerEdge(self):                                 # :349-362
        if 'lin' in self.xscale:
            self.xLowerEdge = np.linspace(self.lower, self.upper, self.binCount + 1)
        else:
            self.xLowerEdge = np.logspace(np.log10(self.lower), np.log10(self.upper), self.binCount + 1)
        self.xWidth = np.diff(self.xLowerEdge)
        self.xMean = (self.xLowerEdge[:-1] + self.xLowerEdge[1:]) / 2.0     # (numpy's mean of two: their sum over 2)

    def calc(self, contribs, paramIndex, fractions):          # :420-439
        self._setXLowerEdge()
        numContribs, dummy, numReps = contribs.shape
        frac, minReq = fractions[self.yweight]
        # _calcBins/_calcBin (:441-469) for every (bin, repetition) at once.  A contribution with
        # edge[b] <= x < edge[b+1] belongs to bin b; numpy.bincount walks its input once, in order, adding weight i to cell
        # index[i] — with the (contribution, repetition) array flattened row-major a cell's members are added in contribution
        # order, the order of the reference's builtin sum() over the masked contributions.
        nb = self.binCount
        par = np.ascontiguousarray(contribs[:, paramIndex, :])
        b = np.searchsorted(self.xLowerEdge, par.ravel(), side='right').reshape(par.shape) - 1
        inb = (b >= 0) & (b < nb) & (par < self.xLowerEdge[-1])
        cell = (b * numReps + np.arange(numReps)[None, :])[inb]
        ncell = nb * numReps
        bins = np.bincount(cell, weights=frac[inb], minlength=ncell)[:ncell].reshape(nb, numReps)
        obsSum = np.bincount(cell, weights=minReq[inb], minlength=ncell)[:ncell].reshape(nb, numReps)
        cnt = np.bincount(cell, minlength=ncell)[:ncell].reshape(nb, numReps).astype(float)
        bins[np.isnan(bins)] = 0.
        with np.errstate(invalid='ignore', divide='ignore'):
            allObs = np.where(cnt > 0, obsSum / cnt, 0.)      # mean of the members' visibility limits
        cdf = np.cumsum(bins, axis=0)                         # _calcCDF :471-479
        top = cdf.max(axis=0) if nb else np.zeros(numReps)
        with np.errstate(invalid='ignore', divide='ignore'):
            cdf = np.where(top[None, :] == 0.0, 0., cdf / top[None, :])
        self._setResults(bins, allObs, cdf)
        self.moments = Moments(contribs, paramIndex, self.xrange, frac)

    def _setResults(self, bins, allObs, cdf):
        nb, numReps = bins.shape
        self.bins = VectorResult(bins)
        self.cdf = VectorResult(cdf)
        finite = np.where(allObs < np.inf, allObs, -np.inf)   # _setObservability :390-402
        best = finite.max(axis=1) if numReps else np.zeros(nb)
        self.observability = np.where(np.isfinite(best), best, 0.)

    def deviceSpec(self, paramIndex):
        """What mcsas_hip_histogram needs to know about this histogram (engine.histogram_device)."""
        self._setXLowerEdge()
        return dict(param_index=paramIndex, yweight=self.yweight, edges=self.xLowerEdge, lower=min(self.xrange), upper=max(self.xrange))

    def setFromDevice(self, res):
        """bins / obs / cdf / moments of every repetition as mcsas_hip_histogram returns them (engine.histogram_device)."""
        self._setResults(np.array(res["bins"]), np.array(res["obs"]), np.array(res["cdf"]))
        self.moments = Moments.fromRepetitions(*[np.array(x) for x in res["moments"]])

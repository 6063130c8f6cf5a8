"""SASData: the three vectors the hot path reads, under the attribute names the reference uses
(dataobj/sasdata.py:51-75, dataobj/datavector.py:46-116).  File parsing, unit conversion, masking
and log-rebinning stay with the reference front-end (out of scope); `fromCsv` only
covers the plain `q; I; sigma` text layout of testdata/quickstartdemo1.csv for demos and tests.

Beam-profile smearing is configured here exactly as in the reference
(`data.config.smearing`, dataobj/sasconfig.py:17-339): this module only prepares the integration
offsets, profile weights and evaluation points `data.locs`; the smeared intensities themselves are
evaluated by the HIP kernels (include/mcsas_hip.h, smear_* fields)."""
from __future__ import annotations

import numpy as np


class DataVector(object):
    def __init__(self, name, data, dataU=None):
        self.name = name
        self.binnedData = np.asarray(data, dtype=float)
        self.binnedDataU = None if dataU is None else np.asarray(dataU, dtype=float)
        self.sanitized = self.binnedData
        self.limit = [float(self.binnedData.min()), float(self.binnedData.max())] if len(self.binnedData) else [0., 0.]


class _Value(object):
    """Configuration parameter with the reference's accessors (`p()`, `p.value()`, `p.setValue(v)`,
    `p.setValueRange((lo, hi))`); numbers are clipped into the value range like Parameter.setValue
    (bases/algorithm/parameter.py:405-414)."""

    def __init__(self, name, default, valueRange=None, onUpdate=None):
        self._name, self._value, self._range, self._onUpdate = name, default, valueRange, onUpdate

    def name(self):
        return self._name

    def value(self):
        return self._value

    __call__ = value

    def valueRange(self):
        return self._range

    def max(self):
        return self._range[1]

    def setValueRange(self, rng):
        self._range = (min(rng), max(rng))
        if not isinstance(self._value, bool):
            self._value = type(self._value)(min(max(self._value, self._range[0]), self._range[1]))

    def setValue(self, v):
        if isinstance(self._value, bool):
            self._value = bool(v)
        else:
            if self._range is not None:
                v = min(max(v, self._range[0]), self._range[1])
            self._value = type(self._value)(v)
        if self._onUpdate is not None:
            self._onUpdate()


class SmearingConfig(object):
    """dataobj/sasconfig.py:17-75: common part of the beam-profile descriptions."""

    def __init__(self):
        self.doSmear = _Value("doSmear", False)
        self.nSteps = _Value("nSteps", 25, (0, 1000))
        self.twoDColl = _Value("twoDColl", False)
        self._qOffset = self._weights = None

    @property
    def qOffset(self):
        return self._qOffset

    @property
    def weights(self):
        return self._weights

    @property
    def prepared(self):
        return self._qOffset, self._weights

    def updateSmearingLimits(self, q):
        pass

    def _offsets(self, lo, hi):
        """Integration offsets: `nSteps` log-spaced points plus zero for slit collimation, mirrored
        ceil(nSteps/2) points plus zero for 2-D (pinhole) collimation (sasconfig.py:132-142)."""
        n = int(self.nSteps())
        if self.twoDColl():
            off = np.logspace(np.log10(lo), np.log10(hi), num=int(np.ceil(n / 2.)))
            return np.concatenate((-off[::-1], [0.], off))
        return np.concatenate(([0.], np.logspace(np.log10(lo), np.log10(hi), num=n)))


class TrapezoidSmearing(SmearingConfig):
    """Trapezoidal beam profile with top width `umbra` and bottom width `penumbra`
    (sasconfig.py:77-184)."""

    def __init__(self):
        super(TrapezoidSmearing, self).__init__()
        self.penumbra = _Value("penumbra", 0., (0., np.inf))
        self.umbra = _Value("umbra", 0., (0., np.inf), onUpdate=self.onUmbraUpdate)

    def onUmbraUpdate(self):                                 # :178-182
        self.penumbra.setValueRange((self.umbra(), self.penumbra.max()))

    def inputValid(self):                                    # :94-96
        return (self.umbra() > 0.) and (self.penumbra() > self.umbra())

    @staticmethod
    def halfTrapzPDF(x, c, d):                               # :104-120
        assert d > 0.
        x = np.abs(x)
        pdf = x * 0.
        pdf[x < c] = 1.
        if d > c:
            slope = (c <= x) & (x < d)
            pdf[slope] = (1. / (d - c)) * (d - x[slope])
        norm = 1. / (d + c)
        return pdf * norm, norm

    def setIntPoints(self, q):                               # :122-149
        xt, xb = self.umbra(), self.penumbra()
        off = self._offsets(q.min() / 5., xb / 2.)
        y, _ = self.halfTrapzPDF(off, xt, xb)
        self._qOffset, self._weights = off, y

    def updateSmearingLimits(self, q):                       # :169-173
        low, high = np.absolute(np.diff(q)).min(), q.max()
        self.umbra.setValueRange((low, 2. * high))
        self.penumbra.setValueRange((low, 2. * high))


class GaussianSmearing(SmearingConfig):
    """Gaussian beam profile; the parameter the reference calls `variance` is used as the standard
    deviation of the profile (sasconfig.py:186-260, scipy.stats.norm.pdf(scale=variance))."""

    def __init__(self):
        super(GaussianSmearing, self).__init__()
        self.variance = _Value("variance", 0., (0., np.inf))

    def inputValid(self):                                    # :198-200
        return self.variance() > 0.

    def setIntPoints(self, q):                               # :209-233
        g = self.variance()
        off = self._offsets(q.min() / 3., 2.5 * g)
        self._qOffset = off
        self._weights = np.exp(-0.5 * (off / g) ** 2) / (g * np.sqrt(2. * np.pi))

    def updateSmearingLimits(self, q):                       # :251-256
        low, high = np.absolute(np.diff(q)).min(), q.max()
        self.variance.setValueRange((low, 2. * high))


class SASConfig(object):
    """The part of dataobj/sasconfig.py:262-360 the Monte-Carlo path reads: `smearing`."""

    def __init__(self, smearing=None):
        self.smearing = smearing if smearing is not None else TrapezoidSmearing()

    def prepareSmearing(self, q):                            # :308-339
        q = np.asarray(q, dtype=float)
        assert q.ndim == 1
        sm = self.smearing
        if sm is None or not sm.inputValid() or not sm.doSmear():
            if sm is not None:
                sm._qOffset = sm._weights = None
            return q
        sm.setIntPoints(q)
        qOffset, weights = sm.prepared
        if not sm.twoDColl():                                # slit collimation
            return np.sqrt(np.add.outer(q ** 2, qOffset ** 2))
        return np.add.outer(q, qOffset)                      # azimuthally averaged 2-D pattern


class SmearArgs(object):
    """What the kernels need of a prepared smearing configuration (engine.HipProblem `smear=`)."""

    def __init__(self, locs, q_offset, weights):
        self.locs, self.q_offset, self.weights = locs, q_offset, weights


class SASData(object):
    def __init__(self, q, intensity, sigma, f_limit=None, title="data", config=None):
        self.title = title
        self.x0 = DataVector("q", q)
        self.f = DataVector("I", intensity, sigma)
        if f_limit is not None:          # limits of the UN-binned intensities (datavector.py:52)
            self.f.limit = [float(f_limit[0]), float(f_limit[1])]
        self.config = config if config is not None else SASConfig()
        if self.config.smearing is not None and self.count > 1:
            self.config.smearing.updateSmearingLimits(self.x0.binnedData)
        self.updateConfig()

    def updateConfig(self):              # dataobj/sasdata.py:161-165: call after changing config.smearing
        self.locs = self.config.prepareSmearing(self.x0.binnedData)

    def smearArgs(self, model):
        """-> SmearArgs when calcIntensity would take the smeared branch for `model`
        (bases/model/sasmodel.py:56-60), else None."""
        sm = self.config.smearing
        if (sm is None or not getattr(model, "canSmear", False) or not sm.doSmear() or not sm.inputValid()):
            return None
        self.updateConfig()
        return SmearArgs(self.locs, sm.qOffset, sm.weights)

    @property
    def q(self):
        return self.x0.binnedData

    @property
    def count(self):
        return len(self.x0.binnedData)

    def sphericalSizeEst(self):          # dataobj/sasdata.py:105 (pi / q limits)
        return np.array([np.pi / self.x0.limit[1], np.pi / self.x0.limit[0]])

    @classmethod
    def fromRaw(cls, q, intensity, sigma=None, nBin=100, fuMin=0.01, title="data", device=-1):
        """Raw SI vectors -> the data object McSAS.analyse reads, the way DataObj prepares it
        (dataobj/dataobj.py): uncertainty floor fuMin*I (_prepareUncertainty, :204-227), points with
        q <= 0 or non-finite entries dropped (the default masks, :239-263), then log-spaced rebinning into
        at most nBin bins (_reBin, :288-345; nBin = 0: none).  Both steps run on the GPU
        (mcsas_hip_prepare_uncertainty, mcsas_hip_rebin)."""
        from . import engine
        q, intensity = np.asarray(q, dtype=float), np.asarray(intensity, dtype=float)
        su = engine.prepare_uncertainty(intensity, sigma, fuMin, device=device)
        ok = np.isfinite(q) & (q > 0.) & np.isfinite(intensity)
        qs, fs, us = q[ok], intensity[ok], su[ok]
        f_limit = [float(fs.min()), float(fs.max())]         # limits of the un-binned intensities (datavector.py:52)
        if nBin and nBin > 0:
            qs, fs, us = engine.rebin(qs, fs, us, int(nBin), device=device)
        return cls(qs, fs, us, f_limit=f_limit, title=title)

    @classmethod
    def fromCsv(cls, filename, q_unit=1e9):
        raw = open(filename, "rb").read().decode("utf-8", "replace").replace("\r", "\n")
        rows = []
        for line in raw.split("\n"):
            parts = [x for x in line.replace(";", " ").replace(",", " ").split() if x]
            try:
                rows.append([float(x) for x in parts[:3]])
            except ValueError:
                continue
        arr = np.array([r for r in rows if len(r) == 3])
        return cls(arr[:, 0] * q_unit, arr[:, 1], arr[:, 2], title=filename)

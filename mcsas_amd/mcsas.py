"""McSAS: drop-in mirror of the reference's algorithm object for the Monte-Carlo path.

Same attributes and calls as `mcsas.mcsas.McSAS` (mcsas/mcsas.py:31-179): `McSAS.factory()()`,
`.data`, `.model`, `.result`, `.stop`, the settings as callable parameters (`numContribs()`,
`maxIterations.value()`, `convergenceCriterion.setValue(...)` ...), `calc()`, `analyse()`,
`histogram()`.  `analyse()` runs every repetition as one MI355X kernel launch; nothing is
evaluated on the CPU.  Plotting, HDF5/pickle output and the GUI hooks stay with the reference.
"""
from __future__ import annotations

import ctypes as C
import logging
import os

import numpy as np

from . import engine
from .parameter import isActiveFitParam
from .scatteringmodels import setup_from_model, host_model_calc


class _Setting(object):
    """Algorithm parameter with the reference's accessors (`p()`, `p.value()`, `p.setValue(v)`)."""

    def __init__(self, name, default, valueRange=None):
        self._name, self._value, self._range = name, default, valueRange

    def name(self):
        return self._name

    def value(self):
        return self._value

    __call__ = value

    def setValue(self, v):
        if self._range is not None and not isinstance(v, bool):
            v = min(max(v, self._range[0]), self._range[1])
        self._value = type(self._value)(v) if not isinstance(self._value, bool) else bool(v)


# mcsas/mcsasparameters.json:2-103
_DEFAULTS = (
    ("numContribs", 300, (1, 1e6)), ("numReps", 10, (1, 1e6)), ("maxIterations", 1e5, (1, 1e100)),
    ("compensationExponent", 0.6666666, None), ("convergenceCriterion", 1.0, (0, np.inf)),
    ("findBackground", True, None), ("positiveBackground", False, None),
    ("startFromMinimum", False, None), ("maxRetries", 5, (1, 100)), ("showIncomplete", False, None),
)


class McSAS(object):
    data = None
    model = None
    result = None

    @classmethod
    def factory(cls):                                        # mcsas.py:143-147
        return cls

    def __init__(self, seed=None, device=-1, wavesPerChain=0, execMode=0):
        """`device`: a HIP device ordinal (-1: the current device) or a list of them — the repetitions are then spread
        over those GPUs by the library (contiguous blocks, mcsas_hip.h: n_devices / devices), results in repetition order
        as if one device had run them; `histogram()` uses the first of the list."""
        for name, default, rng in _DEFAULTS:
            setattr(self, name, _Setting(name, default, rng))
        self.seed = seed
        self.devices = tuple(int(d) for d in device) if isinstance(device, (list, tuple)) else ()
        self.device = self.devices[0] if self.devices else int(device)
        self.wavesPerChain = wavesPerChain
        self.execMode = execMode
        self._stop = C.c_int32(0)
        self.details = None
        self.hostRowWindow = 64          # proposals evaluated ahead per chain for models that exist only as Python (mcsas_hip.h)

    # McSAS.stop is polled once per step in the reference (mcsas.py:357); here the word is
    # forwarded to the running kernel by the library
    @property
    def stop(self):
        return bool(self._stop.value)

    @stop.setter
    def stop(self, flag):
        self._stop.value = 1 if flag else 0

    def calc(self, **kwargs):                                # mcsas.py:149-179
        self.result = []
        self.stop = False
        assert self.data is not None
        if self.model is None:
            raise ValueError("McSAS.model is not set")
        if not self.model.paramCount():
            logging.warning("No parameters to analyse given! Breaking up.")
            return
        self.analyse(replay=kwargs.get("replay"))            # (replay: tests feed the uniform stream the reference consumed)
        if not len(self.result):
            return
        self.histogram()

    def _settings(self, numContribs, numReps):
        seed = self.seed
        if seed is None:                                     # reference: unseeded global MT19937
            seed = int.from_bytes(os.urandom(8), "little")
        return engine.Settings(
            n_contrib=int(numContribs), n_reps=int(numReps), max_iter=self.maxIterations.value(),
            comp_exp=self.compensationExponent(), conv_crit=self.convergenceCriterion(),
            find_background=self.findBackground.value(), positive_background=self.positiveBackground.value(),
            start_from_minimum=self.startFromMinimum(), max_retries=int(self.maxRetries()),
            show_incomplete=self.showIncomplete(), seed=seed, device=self.device, devices=self.devices,
            waves_per_chain=self.wavesPerChain, exec_mode=self.execMode)

    def analyse(self, replay=None):                          # mcsas.py:191-285
        if self.result is None:
            self.result = []
        data, model = self.data, self.model
        if not any(isActiveFitParam(p) for p in model.params()):
            numContribs, numReps = 1, 1                      # mcsas.py:198-199: nothing active, nothing to fit
        else:
            numContribs, numReps = self.numContribs(), self.numReps()
        pr = self._problem(numContribs, numReps, replay)
        if pr["model"].model_id == engine.MODEL_HOST:
            # a model with Python formfactor / volume only: its rows are evaluated here, by the model's own calcIntensity with the
            # set-value-and-restore semantics of ScatteringModel.calc (scatteringmodel.py:86-104); the device does the rest
            c = self.compensationExponent()

            def rows(pset):
                return host_model_calc(model, data, pset, c, want_rows=True)[4]
            res = engine.analyse_host_rows(pr["model"], pr["q"], pr["intensity"], pr["sigma"], pr["st"], rows,
                                           replay=pr["replay"], stop=pr["stop"], window=self.hostRowWindow)
        else:
            res = engine.analyse(pr["model"], pr["q"], pr["intensity"], pr["sigma"], pr["st"],
                                 replay=pr["replay"], stop=pr["stop"], smear=pr["smear"])
        self._store(res, numReps)

    def _problem(self, numContribs=None, numReps=None, replay=None):
        """What analyse() hands to the library for the current data / model / settings (engine.analyse_many takes a list of these)."""
        data, model = self.data, self.model
        if numContribs is None:
            active = any(isActiveFitParam(p) for p in model.params())
            numContribs, numReps = (self.numContribs(), self.numReps()) if active else (1, 1)
        smear = data.smearArgs(model) if hasattr(data, "smearArgs") else None   # sasmodel.py:56-60
        return dict(model=setup_from_model(model, data), q=data.q, intensity=data.f.binnedData, sigma=data.f.binnedDataU,
                    st=self._settings(numContribs, numReps), replay=replay, stop=self._stop, smear=smear)

    def _store(self, res, numReps=None):
        """The second half of analyse() (mcsas.py:221-285): warnings, active values, the result dictionary."""
        data, model = self.data, self.model
        if self.result is None:
            self.result = []
        if numReps is None:
            numReps = res.contribs.shape[2]
        self.details = res
        if (res.converged == 0).any():                       # mcsas.py:221-230 / :240-245
            if self.stop:
                logging.warning("Stop button pressed, exiting...")
            else:
                logging.warning("Could not reach optimization criterion within {0} attempts, exiting..."
                                .format(self.maxRetries() + 2))
            if not self.showIncomplete():
                return
        contribs = res.contribs
        for nr in range(numReps):                            # mcsas.py:434-437
            for idx, param in enumerate(model.activeParams()):
                param.setActiveVal(contribs[:, idx, nr].copy(), index=nr)
        meas = res.fit.reshape(1, data.count, numReps)       # contribMeasVal, mcsas.py:210
        ddof = 1 if numReps > 1 else 0
        self.result.append(dict(
            contribs=contribs,
            fitMeasValMean=meas.mean(axis=2), fitMeasValStd=meas.std(axis=2),
            fitX0=data.x0.binnedData, dataX0=data.x0.binnedData,
            dataMean=data.f.binnedData, dataStd=data.f.binnedDataU,
            scaling=(res.scaling.mean(), res.scaling.std(ddof=ddof)),
            background=(res.background.mean(), res.background.std(ddof=ddof)),
            times=res.seconds, numIter=res.num_iter.astype(float).mean()))

    def histogram(self, contribs=None):                      # mcsas.py:445-615
        if not isinstance(self.result, list) or not len(self.result):
            logging.info("There are no results to histogram, breaking up.")
            return
        if contribs is None:
            contribs = self.result[0]['contribs']
        if not all(np.array(contribs.shape, dtype=bool)):
            return
        numContribs, dummy, numReps = contribs.shape
        data, model = self.data, self.model
        setup = setup_from_model(model, data)
        smear = data.smearArgs(model) if hasattr(data, "smearArgs") else None
        c = self.compensationExponent()
        sig = np.array(data.f.binnedDataU, dtype=float)
        if setup.model_id == engine.MODEL_HOST:
            # rows by the model's own calcIntensity (:552, :577-578), the fit on the device (:559), the visibility limits in numpy (:575-590)
            scalingFactors = np.zeros((2, numReps))
            vsets = np.zeros((numContribs, numReps)); wsets = np.zeros((numContribs, numReps)); ssets = np.zeros((numContribs, numReps))
            mv = np.zeros((numContribs, numReps))
            for ri in range(numReps):
                cum, vsets[:, ri], wsets[:, ri], ssets[:, ri], rows = host_model_calc(model, data, contribs[:, :, ri], c, want_rows=True)
                sc, _, _ = engine.bgfit(data.f.binnedData, sig, cum, self.findBackground.value(), self.positiveBackground.value(),
                                        num_params=model.activeParamCount(), device=self.device)
                scalingFactors[:, ri] = sc
                vf_r = wsets[:, ri] * sc[0] / vsets[:, ri]
                for ci in range(numContribs):
                    part = sc[0] * rows[ci]
                    nz = part != 0.
                    mv[ci, ri] = (sig[nz] * vf_r[ci] / part[nz]).min()
            self._histogram_tail(contribs, scalingFactors, vsets, wsets, ssets, mv)
            return
        # Everything on the device in one call (mcsas_hip_histogram): model.calc, the scale / background fit, the visibility
        # limits, the fractions, and per configured histogram the bins, CDF and moments of every repetition — their mean / std over
        # the repetitions is taken here.  (More than engine.HISTOGRAM_MAX_CONTRIBS contributions: the two-step path below.)
        if numContribs <= engine.HISTOGRAM_MAX_CONTRIBS:
            todo = [(pi, h) for pi, param in enumerate(model.activeParams()) for h in param.histograms()]
            specs = [h.deviceSpec(pi) for pi, h in todo]
            scalingFactors, fractions, res = engine.histogram_device(
                setup, data.q, data.f.binnedData, sig, contribs, c, specs, self.findBackground.value(),
                self.positiveBackground.value(), device=self.device, smear=smear)
            self.result[0]['scalingFactors'] = scalingFactors
            self.fractions = fractions
            for (pi, h), r in zip(todo, res):
                h.setFromDevice(r)
            return
        # model.calc, the scale/background fit and the N single-row visibility limits of every repetition
        # (:552, :559, :575-590): one library call for all of them
        scalingFactors, vsets, wsets, ssets, mv = engine.histogram_prep(
            setup, data.q, data.f.binnedData, sig, contribs, c, self.findBackground.value(),
            self.positiveBackground.value(), device=self.device, smear=smear)
        self._histogram_tail(contribs, scalingFactors, vsets, wsets, ssets, mv)

    def _histogram_tail(self, contribs, scalingFactors, vsets, wsets, ssets, mv):
        """mcsas.py:561-615 from the per-contribution volumes / weights / surfaces and visibility limits of every repetition."""
        model = self.model
        vf = wsets * scalingFactors[0][None, :] / vsets          # modeldata.py:57-61
        nf = vf / vsets
        qf = vf * vsets
        sf = nf * ssets
        mn = mv / vsets                                      # :591-594
        mq = mn * mv * mv
        ms = mn * ssets
        # normalisation per repetition (:596-604); cumsum adds in the order of the reference's builtin sum()
        for frac, lim in ((nf, mn), (qf, mq), (sf, ms)):
            tot = np.cumsum(frac, axis=0)[-1]
            nz = tot != 0
            frac[:, nz] /= tot[nz]; lim[:, nz] /= tot[nz]
        fractions = dict(vol=(vf, mv), num=(nf, mn), int=(qf, mq), surf=(sf, ms))
        self.result[0]['scalingFactors'] = scalingFactors
        self.fractions = fractions
        for paramIndex, param in enumerate(model.activeParams()):
            param.histograms().calc(contribs, paramIndex, fractions)

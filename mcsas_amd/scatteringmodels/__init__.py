"""Scattering-model plugins: host-side mirror of the reference's ScatteringModel / SASModel API
(bases/model/scatteringmodel.py:14-127, bases/model/sasmodel.py:11-79, bases/model/modeldata.py)
for the four models on the hot path.  A model here only *declares* its parameters; every
evaluation (`calc`) runs in libmcsas_hip.so.
"""
from __future__ import annotations

import numpy as np

from .. import engine
from ..parameter import (Parameter, FitParameter, RandomUniform, RandomExponential, isActiveFitParam,
                         generator_kind)

NM = 1e-9
SLD_A2 = 1e20          # Å⁻² -> m⁻²  (utils/units.py SLD)


class SASModelData(object):
    """bases/model/modeldata.py:4-64."""

    def __init__(self, cumInt, vset, wset, sset, numParams):
        self._cumInt = np.asarray(cumInt).flatten(); self._vset = np.asarray(vset).flatten()
        self._wset = np.asarray(wset).flatten(); self._sset = np.asarray(sset).flatten()
        self._numParams = abs(numParams)

    cumInt = property(lambda self: self._cumInt)
    chisqrInt = property(lambda self: self._cumInt)          # modeldata.py:19-24
    vset = property(lambda self: self._vset)
    wset = property(lambda self: self._wset)
    sset = property(lambda self: self._sset)
    numParams = property(lambda self: self._numParams)

    def volumeFraction(self, scaling):                       # modeldata.py:57-61
        return (self.wset * scaling / self.vset).flatten()


class ScatteringModel(object):
    """Declares `parameters` (a tuple of Parameter/FitParameter factories) like the reference's
    model classes; instances get one attribute per parameter."""
    shortName = None
    model_id = None
    parameters = ()

    def __init__(self):
        self._params = []
        for make in self.parameters:
            p = make()
            setattr(self, p.name(), p)
            self._params.append(p)

    @classmethod
    def name(cls):
        return cls.shortName or cls.__name__

    def params(self):
        return tuple(self._params)

    def paramCount(self):
        return len(self._params)

    def activeParams(self):
        return tuple(p for p in self._params if isActiveFitParam(p))

    def activeParamCount(self):
        return len(self.activeParams())

    def modelDataType(self):
        return SASModelData

    def getModelData(self, cumInt, vset, wset, sset):        # scatteringmodel.py:107-109
        return self.modelDataType()(np.asarray(cumInt).flatten(), vset, wset, sset, self.activeParamCount())

    def generateParameters(self, count=1):                   # scatteringmodel.py:117-127
        lst = np.zeros((count, self.activeParamCount()))
        for idx, param in enumerate(self.activeParams()):
            lst[:, idx] = param.generate(count=count)
        return lst

    def setup(self, data=None) -> engine.ModelSetup:
        return setup_from_model(self, data)

    def calc(self, data, pset, compensationExponent=None):   # scatteringmodel.py:79-105, on the GPU
        if is_host_model(self):                              # a model of the user's own with Python formfactor / volume only
            return self.getModelData(*host_model_calc(self, data, np.asarray(pset, dtype=float), compensationExponent)[:4])
        q = data.q if hasattr(data, "q") else np.asarray(data)
        smear = data.smearArgs(self) if hasattr(data, "smearArgs") else None    # sasmodel.py:56-60
        cum, v, w, s = engine.model_calc(self.setup(), q, pset, compensationExponent, smear=smear)
        return self.getModelData(cum, v, w, s)

    # ---- what a model of the user's own supplies in Python (bases/model/scatteringmodel.py:15-58); the built-in models never
    # call these: their arithmetic is in csrc/models.h
    def volume(self):
        raise NotImplementedError

    def absVolume(self):                                     # scatteringmodel.py:22-25
        return self.volume()

    def surface(self):                                       # :53-57
        return 0

    def formfactor(self, dataset):
        raise NotImplementedError


class SASModel(ScatteringModel):
    canSmear = False
    compensationExponent = None

    def getQ(self, dataset):                                 # sasmodel.py:27-35
        return dataset if isinstance(dataset, np.ndarray) else dataset.q

    def weight(self):                                        # sasmodel.py:37-44
        return self.volume() ** (2 * self.compensationExponent)

    def calcIntensity(self, data, compensationExponent=None):
        """sasmodel.py:46-79 for a model with Python formfactor / volume: (it, v, w, s) of the CURRENT parameter values."""
        self.compensationExponent = compensationExponent
        v = self.absVolume()
        w = self.weight()
        s = self.surface()
        sm = getattr(getattr(data, "config", None), "smearing", None)
        if sm is not None and self.canSmear and sm.doSmear() and sm.inputValid():
            ff = self.formfactor(data.locs)
            qOffset, weightFunc = sm.prepared
            it = 2 * _trapz(ff ** 2 * w * weightFunc, x=qOffset, axis=1)
        else:
            ff = self.formfactor(data)
            it = ff ** 2 * w
        return it, v, w, s


_trapz = getattr(np, "trapezoid", None) or np.trapz


def _fp(*a, **k):
    return lambda: FitParameter(*a, **k)


def _p(*a, **k):
    return lambda: Parameter(*a, **k)


class Sphere(SASModel):
    """models/sphere.py:12-65."""
    shortName = "Sphere"
    model_id = engine.MODEL_SPHERE
    canSmear = True
    parameters = (
        _fp("radius", 10. * NM, displayName="Sphere radius", valueRange=(0., np.inf),
            activeRange=(1. * NM, 1000. * NM), generator=RandomUniform),
        _p("sld", 1e-6 * SLD_A2, displayName="scattering length density difference", valueRange=(0., np.inf)),
    )

    def __init__(self):
        super().__init__()
        self.radius.setActive(True)


class CylindersIsotropic(SASModel):
    """models/cylindersisotropic.py:17-103."""
    shortName = "SASfit Isotropic Cylinders"
    model_id = engine.MODEL_CYL_ISO
    parameters = (
        _fp("radius", 1. * NM, displayName="Cylinder Radius", generator=RandomExponential,
            valueRange=(0.1 * NM, np.inf)),
        _p("useAspect", True, displayName="Use aspect ratio (checked) or length "),
        _fp("length", 10. * NM, displayName="Length L of the Cylinder", generator=RandomExponential,
            valueRange=(0.1 * NM, 1e10 * NM)),
        _fp("aspect", 10.0, displayName="Aspect ratio of the Cylinder", generator=RandomExponential,
            valueRange=(1e-3, 1e3)),
        _p("intDiv", 100., displayName="Orientation Integration Divisions", valueRange=(1, 1e4)),
        _p("sld", 1e-6 * SLD_A2, displayName="Scattering length density difference", valueRange=(0, np.inf)),
    )

    def __init__(self):
        super().__init__()
        self.radius.setActive(True)


class EllipsoidalCoreShell(SASModel):
    """models/ellipsoidalcoreshell.py:14-99."""
    shortName = "Core-Shell Ellipsoid"
    model_id = engine.MODEL_ELL_CS
    parameters = (
        _fp("a", 1. * NM, displayName="Principal Core Radius", generator=RandomExponential,
            valueRange=(0., np.inf), activeRange=(0.1 * NM, 1e3 * NM)),
        _fp("b", 10. * NM, displayName="Equatorial Core Radius", generator=RandomExponential,
            valueRange=(0., np.inf), activeRange=(1.0 * NM, 1e4 * NM)),
        _fp("t", 1. * NM, displayName="Thickness of Shell", generator=RandomExponential,
            valueRange=(0., np.inf), activeRange=(0.1 * NM, 1e3 * NM)),
        _p("eta_c", 3.15e-6 * SLD_A2, displayName="Core SLD", valueRange=(0, np.inf)),
        _p("eta_s", 2.53e-6 * SLD_A2, displayName="Shell SLD", valueRange=(0, np.inf)),
        _p("eta_sol", 0., displayName="Solvent SLD", valueRange=(0, np.inf)),
        _p("intDiv", 100, displayName="Orientation Integration Divisions", valueRange=(0, 1e4)),
    )

    def __init__(self):
        super().__init__()
        self.a.setActive(True)


class Kholodenko(SASModel):
    """models/kholodenko.py:51-96."""
    shortName = "Kholodenko Worm"
    model_id = engine.MODEL_KHOLODENKO
    parameters = (
        _fp("radius", 1. * NM, displayName="Radius", generator=RandomExponential,
            valueRange=(0., np.inf), activeRange=(1 * NM, 5 * NM)),
        _fp("lenKuhn", 1. * NM, displayName="kuhn length", generator=RandomUniform,
            valueRange=(0., np.inf), activeRange=(10 * NM, 50 * NM)),
        _fp("lenContour", 2. * NM, displayName="contour length", generator=RandomUniform,
            valueRange=(0., np.inf), activeRange=(100 * NM, 1000 * NM)),
    )

    def __init__(self):
        super().__init__()
        self.radius.setActive(True)
        self.lenKuhn.setActive(True)
        self.lenContour.setActive(True)


class EllipsoidsIsotropic(SASModel):
    """models/ellipsoidsisotropic.py:18-86."""
    shortName = "Isotropic Ellipsoids"
    model_id = engine.MODEL_ELL_ISO
    parameters = (
        _fp("a", 1. * NM, displayName="Radius of semi-axes a, b", generator=RandomExponential,
            valueRange=(0.1 * NM, 1e10 * NM), activeRange=(0.1 * NM, 1e3 * NM)),
        _p("useAspect", True, displayName="Use aspect ratio (checked) or length to define c-axis"),
        _fp("c", 10. * NM, displayName="Radius of semi-axes c", generator=RandomExponential,
            valueRange=(0.1 * NM, 1e10 * NM), activeRange=(1. * NM, 1e4 * NM)),
        _fp("aspect", 10.0, displayName="aspect ratio of semi-axes c to a, b", generator=RandomExponential,
            valueRange=(1e-3, 1e3)),
        _p("intDiv", 100, displayName="Orientation Integration Divisions", valueRange=(0, 1e4)),
        _p("sld", 1e-6 * SLD_A2, displayName="Scattering length density difference", valueRange=(0, 1e-2 * SLD_A2)),
    )

    def __init__(self):
        super().__init__()
        self.a.setActive(True)


class SphericalCoreShell(SASModel):
    """models/sphericalcoreshell.py:14-79."""
    shortName = "Core-Shell Sphere"
    model_id = engine.MODEL_SPH_CS
    parameters = (
        _fp("radius", 1. * NM, displayName="Core Radius", generator=RandomExponential,
            valueRange=(0., np.inf), activeRange=(0.1 * NM, 1e3 * NM)),
        _fp("t", 1. * NM, displayName="Thickness of Shell", generator=RandomExponential,
            valueRange=(0., np.inf), activeRange=(0.1 * NM, 1e3 * NM)),
        _p("eta_c", 3.16e-6 * SLD_A2, displayName="Core SLD", valueRange=(0, np.inf)),
        _p("eta_s", 2.53e-6 * SLD_A2, displayName="Shell SLD", valueRange=(0, np.inf)),
        _p("eta_sol", 0., displayName="Solvent SLD", valueRange=(0, np.inf)),
    )

    def __init__(self):
        super().__init__()
        self.radius.setActive(True)


class GaussianChain(SASModel):
    """models/gaussianchain.py:14-76."""
    shortName = "Gaussian Chain"
    model_id = engine.MODEL_GAUSS_CHAIN
    parameters = (
        _fp("rg", 1. * NM, displayName="radius of gyration, Rg", generator=RandomExponential,
            valueRange=(0., np.inf), activeRange=(1.0 * NM, 1e2 * NM)),
        _fp("bp", 100. * NM, displayName="scattering length of the polymer", generator=RandomUniform,
            valueRange=(0., np.inf), activeRange=(0.1 * NM, 1e3 * NM)),
        _fp("etas", 1e-6 * SLD_A2, displayName="scattering length density of the solvent", generator=RandomUniform,
            valueRange=(0., np.inf), activeRange=(0.1 * SLD_A2, 10. * SLD_A2)),
        _fp("k", 1.0, displayName="volumetric scaling factor of Rg", generator=RandomUniform,
            valueRange=(0., np.inf), activeRange=(0.1, 10.)),
    )

    def __init__(self):
        super().__init__()
        self.rg.setActive(True)


class LMADenseSphere(SASModel):
    """models/lmadensesphere.py:14-108."""
    shortName = "LMADenseSphere"
    model_id = engine.MODEL_LMA_SPHERE
    canSmear = True
    parameters = (
        _fp("radius", 1. * NM, displayName="Sphere radius", valueRange=(0., np.inf), generator=RandomUniform),
        _fp("volFrac", 0.10, displayName="Volume fraction of spheres", valueRange=(0.001e-2, 1.0), generator=RandomUniform),
        _p("mf", -1., displayName="standoff multiplier (-1 = auto)", valueRange=(-1., 1.e6)),
        _p("sld", 1e-6 * SLD_A2, displayName="Scattering length density difference", valueRange=(0., np.inf)),
    )

    def __init__(self):
        super().__init__()
        self.radius.setActive(True)


def _cyl_radially_isotropic_source(psi_range):
    """HIP text of models/cylindersradiallyisotropic.py:49-81 for the run-time plug-in path (INTEGRATION.md "Models of the user's
    own"): the in-plane orientation average is a loop inside the form factor, over psi = numpy.linspace(*psiAngle.valueRange(),
    psiAngleDivisions) — the range is the PARAMETER'S valueRange at the time of the call, like the reference, hence part of the text."""
    lo, hi = (repr(float(x)) for x in psi_range)
    return r"""
// models/cylindersradiallyisotropic.py:49-81; p = (radius, aspect, psiAngle, psiAngleDivisions, sld)
#define MCSAS_PLUGIN_ROW_CLASS 1
__device__ double mcsas_plugin_volume(const double *p) { return mcsas::PI * (p[0] * p[0]) * (2. * p[0] * p[1]); }
__device__ double mcsas_plugin_absvolume(const double *p) { return mcsas_plugin_volume(p) * (p[4] * p[4]); }
__device__ double mcsas_plugin_surface(const double *p) { return 0.; }
__device__ double mcsas_plugin_formfactor(double q, const double *p) {
    const int K = (int)p[3];
    const double lo = %s, hi = %s, step = (hi - lo) / (double)(K > 1 ? K - 1 : 1);
    double sum = 0.;
    for (int k = 0; k < K; ++k) {
        const double psi = (k == K - 1 && K > 1 ? hi : lo + step * (double)k) - p[2];     // numpy.linspace, minus the rotation
        double sa, ca;
        sincos(psi, &sa, &ca);
        const double xr = fabs(q * (p[0] * sa)), xl = q * (p[0] * p[1] * ca);           // J1(x) / x is even
        const double f = 2. * mcsas::j1_fast(xr) / xr * sin(xl) / xl;
        sum += f * f;
    }
    return sqrt(sum / (double)K);
}
""" % (lo, hi)


class CylindersRadiallyIsotropic(SASModel):
    """models/cylindersradiallyisotropic.py:14-83 ("completed but not verified" there; of the reference's three cylinder variants
    outside its verified set the one that runs as written).  No built-in kernel id: its form factor reaches the library as HIP
    text (`hipSource`) and runs the row-queue kernels compiled at run time, like a model of the user's own."""
    shortName = "Radially (in-plane) isotropic cylinders"
    model_id = None
    parameters = (
        _fp("radius", 1. * NM, displayName="Cylinder radius", generator=RandomExponential,
            valueRange=(0.1 * NM, np.inf), activeRange=(0.1 * NM, 1e3 * NM)),
        _fp("aspect", 10.0, displayName="Aspect ratio L/(2R) of the cylinder", generator=RandomUniform,
            valueRange=(0.1, np.inf), activeRange=(1.0, 20.)),
        _fp("psiAngle", 0.17, displayName="in-plane cylinder rotation", generator=RandomUniform,
            valueRange=(0.01, 2 * np.pi + 0.01)),
        _p("psiAngleDivisions", 303., displayName="in-plane angle divisions", valueRange=(1, np.inf)),
        _p("sld", 1e-6 * SLD_A2, displayName="scattering length density difference", valueRange=(0., np.inf)),
    )

    def __init__(self):
        super().__init__()
        self.radius.setActive(True)
        self.aspect.setActive(False)
        self.psiAngle.setActive(True)

    @property
    def hipSource(self):
        return _cyl_radially_isotropic_source(self.psiAngle.valueRange())


# reference class name -> HIP text of the models that ship as run-time plug-ins (a reference object of that class flattens like ours)
SHIPPED_PLUGINS = {"CylindersRadiallyIsotropic": lambda model: _cyl_radially_isotropic_source(model.psiAngle.valueRange())}

# reference class name -> kernel id (FindModels walks models/*.py, utils/findmodels.py:120-186)
MODEL_IDS = {"Sphere": engine.MODEL_SPHERE, "CylindersIsotropic": engine.MODEL_CYL_ISO,
             "EllipsoidalCoreShell": engine.MODEL_ELL_CS, "Kholodenko": engine.MODEL_KHOLODENKO,
             "EllipsoidsIsotropic": engine.MODEL_ELL_ISO, "SphericalCoreShell": engine.MODEL_SPH_CS,
             "GaussianChain": engine.MODEL_GAUSS_CHAIN, "LMADenseSphere": engine.MODEL_LMA_SPHERE}


def is_host_model(model) -> bool:
    """A model without a kernel id, without a shipped plug-in text and without `hipSource`, but with the reference's Python
    contract (calcIntensity, i.e. formfactor + volume: bases/model/sasmodel.py:46-79): its rows are evaluated by the host."""
    if getattr(model, "model_id", None) is not None or type(model).__name__ in MODEL_IDS:
        return False
    if isinstance(getattr(model, "hipSource", None), str) or type(model).__name__ in SHIPPED_PLUGINS:
        return False
    if isinstance(model, ScatteringModel):                  # our mirror base: the subclass must bring formfactor AND volume
        return type(model).formfactor is not ScatteringModel.formfactor and type(model).volume is not ScatteringModel.volume
    return callable(getattr(model, "calcIntensity", None))  # the reference's own classes: ABCMeta enforces the rest


def host_model_calc(model, data, pset, compensationExponent, want_rows=False):
    """ScatteringModel.calc (bases/model/scatteringmodel.py:79-105) for a host model, ours or the reference's own object: remember the
    active parameters' values, per parameter set p.setValue(v) (which clips into the valueRange) and calcIntensity, rows summed
    in order, values restored.  -> (cumInt, vset, wset, sset, rows or None)."""
    params = model.activeParams()
    old = [p() for p in params]
    pset = np.asarray(pset, dtype=float).reshape(-1, len(params))
    cum = None
    vset = np.zeros(len(pset)); wset = np.zeros(len(pset)); sset = np.zeros(len(pset))
    rows = [] if want_rows else None
    try:
        for i, row in enumerate(pset):
            for p, v in zip(params, row):
                p.setValue(float(v))
            it, vset[i], wset[i], sset[i] = model.calcIntensity(data, compensationExponent=compensationExponent)
            it = np.asarray(it, dtype=float).flatten()
            cum = it.copy() if cum is None else cum + it
            if want_rows:
                rows.append(it)
    finally:
        for p, v in zip(params, old):
            p.setValue(v)
    if cum is None:
        cum = np.zeros(0)
    return cum, vset, wset, sset, (np.array(rows) if want_rows else None)


def _as_float(v):
    try:
        return float(v)
    except (TypeError, ValueError):
        return 0.0


def setup_from_model(model, data=None) -> engine.ModelSetup:
    """Flattens a configured model instance — ours or the reference's own (duck-typed through
    params()/name()/value()/isActive()/activeRange()/valueRange()/generator()) — into the
    ModelSetup the C ABI takes.  Unknown plugins have no device implementation: loud error."""
    mid = getattr(model, "model_id", None)
    if mid is None:
        mid = MODEL_IDS.get(type(model).__name__)
    if mid is None and not isinstance(getattr(model, "hipSource", None), str) and type(model).__name__ in SHIPPED_PLUGINS:
        mid = engine.compile_plugin(SHIPPED_PLUGINS[type(model).__name__](model))
    if mid is None and isinstance(getattr(model, "hipSource", None), str):
        # a model of the user's own (the reference: any models/*.py, utils/findmodels.py:120-186): its form factor as HIP
        # source text, compiled at run time into the wave-per-chain kernel (engine.compile_plugin)
        mid = engine.compile_plugin(model.hipSource)
    if mid is None and is_host_model(model):
        # Python formfactor / volume only (the reference's plug-in contract as it stands): the chains run with host-evaluated
        # rows (mcsas_hip_analyse_host_rows); everything but the rows stays on the device
        mid = engine.MODEL_HOST
    if mid is None:
        raise NotImplementedError("model %s has no HIP kernel (built in: %s), no `hipSource` text and no Python calcIntensity "
                                  "(formfactor + volume) either: see INTEGRATION.md" % (type(model).__name__, ", ".join(MODEL_IDS)))
    params = list(model.params())
    if len(params) > engine.MAX_PARAMS and mid != engine.MODEL_HOST:
        raise NotImplementedError("model %s has %d parameters, the C ABI carries %d" % (type(model).__name__, len(params), engine.MAX_PARAMS))
    # (a model whose rows the host evaluates keeps its parameter vector to itself: only the ACTIVE parameters' generator ranges
    # travel, so it may declare any number of parameters of any type)
    values = np.array([_as_float(p()) for p in params[:engine.MAX_PARAMS]], dtype=float)
    active, lo, hi, kind, clo, chi, start = [], [], [], [], [], [], []
    for i, p in enumerate(params):
        if not isActiveFitParam(p):
            continue
        vr = p.valueRange()
        ar = p.activeRange()
        active.append(i)
        lo.append(max(vr[0], min(ar))); hi.append(min(vr[1], max(ar)))     # parameter.py:66-84
        kind.append(generator_kind(p.generator()))
        clo.append(vr[0]); chi.append(vr[1])
        mb = min(ar)                                                          # mcsas.py:311-315
        if mb == 0 and data is not None:
            mb = np.pi / data.x0.limit[1]
        start.append(mb * .5)
    return engine.ModelSetup(mid, values, tuple(active), np.array(lo), np.array(hi), tuple(kind),
                             np.array(clo), np.array(chi), np.array(start))

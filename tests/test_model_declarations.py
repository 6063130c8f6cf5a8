"""The mirror model classes (mcsas_amd/scatteringmodels) against what the reference's own classes declare.

(1) tests/golden/g10_param_decls.json — written by oracle/make_golden.py from the imported reference: per model
    class every parameter's name, default, valueRange, activeRange, generator and active flag
    (models/*.py, utils/parameter.py:577-743, bases/algorithm/parameter.py:420-433).  Runs everywhere.
(2) In the build container only (skipped when /root/reference is absent, e.g. on the GPU box): the reference's
    OWN model instances, configured through their own setters, go through mcsas_amd's duck-typed
    `setup_from_model` — the binding INTEGRATION.md describes — and must flatten to the same mcsas_problem
    fields as the mirror classes configured the same way (bases/model/scatteringmodel.py:117-127)."""
import json
import os
import sys

import numpy as np
import pytest

import mcsas_amd
from mcsas_amd import scatteringmodels as SM

HERE = os.path.dirname(os.path.abspath(__file__))
DECLS = json.load(open(os.path.join(HERE, "golden", "g10_param_decls.json")))


@pytest.mark.parametrize("cls", sorted(DECLS))
def test_mirror_declarations_equal_the_reference(cls):
    ref = DECLS[cls]
    m = getattr(SM, cls)()
    assert m.shortName == ref["shortName"]
    assert bool(getattr(m, "canSmear", False)) == ref["canSmear"]
    assert [p.name() for p in m.params()] == [e["name"] for e in ref["params"]]
    assert [p.name() for p in m.activeParams()] == ref["activeParams"]
    for e, p in zip(ref["params"], m.params()):
        assert float(p()) == e["value"], (cls, e["name"])
        if e["valueRange"] is not None:
            assert [float(v) for v in p.valueRange()] == e["valueRange"], (cls, e["name"])
        assert hasattr(p, "activeRange") == e["fit"]
        if e["fit"]:
            ar = p.activeRange()
            assert bool(p.isActive()) == e["active"], (cls, e["name"])
            assert [float(min(ar)), float(max(ar))] == e["activeRange"], (cls, e["name"])
            assert p.generator().__name__ == e["generator"], (cls, e["name"])
    # what the kernels see: same model id table as the reference's class names (or, no built-in kernel: the class ships its HIP text)
    assert SM.MODEL_IDS.get(cls) == m.model_id and (cls in SM.MODEL_IDS or (cls in SM.SHIPPED_PLUGINS and isinstance(m.hipSource, str)))


# ------------------------------------------------------------------------------------------------------------
REF_SRC = "/root/reference/src"


def _import_reference():
    """The reference package with the three third-party import stand-ins of SURVEY.md Appendix A."""
    if not os.path.isdir(REF_SRC):
        pytest.skip("reference not present (it never travels to the GPU box)")
    import tempfile
    d = tempfile.mkdtemp(prefix="mcsas_shim_")
    os.makedirs(os.path.join(d, "future"))
    open(os.path.join(d, "future", "__init__.py"), "w").close()
    with open(os.path.join(d, "future", "standard_library.py"), "w") as f:
        f.write("def install_aliases(): pass\n")
    with open(os.path.join(d, "future", "utils.py"), "w") as f:
        f.write("def with_metaclass(meta, *bases):\n"
                "    class metaclass(type):\n"
                "        def __new__(cls, name, this_bases, d): return meta(name, bases, d)\n"
                "        @classmethod\n"
                "        def __prepare__(cls, name, this_bases): return meta.__prepare__(name, bases)\n"
                "    return type.__new__(metaclass, 'temporary_class', (), {})\n")
    with open(os.path.join(d, "QtWidgets.py"), "w") as f:
        f.write("class QApplication(object):\n"
                "    @staticmethod\n"
                "    def processEvents(): pass\n"
                "    @staticmethod\n"
                "    def translate(ctx, s): return s\n")
    sys.dont_write_bytecode = True
    sys.path[:0] = [d, REF_SRC]
    import logging
    logging.disable(logging.WARNING)
    import mcsas.models.sphere                      # noqa: F401  (ordinary import errors fail the test loudly)
    return d


CONFIGS = {
    # class name -> (active parameters with their active ranges, fixed values, generator overrides)
    "Sphere": (dict(radius=(2e-9, 3e-7)), dict(sld=2.5e14), {}),
    "CylindersIsotropic": (dict(radius=(1e-9, 1e-7), aspect=(0.5, 20.0)), dict(intDiv=100, sld=1e14), {}),
    "EllipsoidalCoreShell": (dict(a=(1e-9, 1e-7), b=(2e-9, 2e-7), t=(2e-10, 1e-8)), dict(eta_c=3e14, eta_s=2e14), {}),
    "Kholodenko": (dict(radius=(1e-9, 4e-9), lenKuhn=(1.5e-8, 4e-8), lenContour=(2e-7, 9e-7)), {}, {}),
    "EllipsoidsIsotropic": (dict(a=(1e-9, 1e-7), aspect=(0.3, 8.0)), dict(sld=1e14), {}),
    "SphericalCoreShell": (dict(radius=(1e-9, 1e-7), t=(5e-10, 2e-8)), dict(eta_sol=1e13), {}),
    "GaussianChain": (dict(rg=(1e-9, 1e-7), bp=(1e-9, 1e-7)), dict(k=1.5), dict(bp="RandomExponential2")),
    "LMADenseSphere": (dict(radius=(2e-9, 2e-7), volFrac=(0.01, 0.4)), dict(mf=-1.0), dict(radius="RandomExponential3")),
}


def _configure(m, gens, active, fixed, gen_over):
    for p in m.params():
        if hasattr(p, "setActive"):
            p.setActive(p.name() in active)
    if type(m).__name__ == "CylindersIsotropic" and hasattr(m.intDiv, "setValue"):
        m.intDiv.setValue(101)                       # SURVEY 8c gotcha 5: leaves an int behind on numpy >= 2
    for k, v in fixed.items():
        getattr(m, k).setValue(v)
    for k, r in active.items():
        getattr(m, k).setActiveRange(r)
    for k, g in gen_over.items():
        getattr(m, k).setGenerator(gens[g])
    return m


@pytest.mark.parametrize("cls", sorted(CONFIGS))
def test_reference_model_objects_flatten_like_the_mirror(cls):
    _import_reference()
    import importlib
    from mcsas.bases.algorithm import numbergenerator as refgen
    modname = {"Sphere": "sphere", "CylindersIsotropic": "cylindersisotropic", "EllipsoidalCoreShell": "ellipsoidalcoreshell",
               "Kholodenko": "kholodenko", "EllipsoidsIsotropic": "ellipsoidsisotropic", "SphericalCoreShell": "sphericalcoreshell",
               "GaussianChain": "gaussianchain", "LMADenseSphere": "lmadensesphere"}[cls]
    RefCls = getattr(importlib.import_module("mcsas.models." + modname), cls)
    active, fixed, gen_over = CONFIGS[cls]
    ref_gens = {n: getattr(refgen, n) for n in ("RandomUniform", "RandomExponential", "RandomExponential2", "RandomExponential3")}
    our_gens = {n: getattr(mcsas_amd, n) for n in ref_gens}
    ref_model = _configure(RefCls(), ref_gens, active, fixed, gen_over)
    our_model = _configure(getattr(SM, cls)(), our_gens, active, fixed, gen_over)

    class X0:                                           # startFromMinimum needs data.x0.limit (mcsas.py:311-315)
        limit = [1e7, 3e9]

    class D:
        x0 = X0()
    a = SM.setup_from_model(ref_model, D())            # the reference's own instance, duck-typed
    b = our_model.setup(D())
    assert a.model_id == b.model_id and a.active_index == b.active_index and a.gen_kind == b.gen_kind
    for f in ("params", "gen_lo", "gen_hi", "clip_lo", "clip_hi", "start_value"):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg="%s.%s" % (cls, f))
    # and the reference's generateParameters draws inside the range the kernels are given
    np.random.seed(5)
    g = np.asarray(ref_model.generateParameters(64), dtype=float)
    assert (g >= a.gen_lo).all() and (g <= a.gen_hi).all()


def test_a_reference_model_that_exists_only_as_python_runs_through_host_rows():
    """A model of the user's own written against the REFERENCE's plug-in API — a SASModel subclass with numpy formfactor() /
    volume() / absVolume() and nothing else (bases/model/scatteringmodel.py:16-58, sasmodel.py:37-79; here the Guinier law of a
    sphere, which no built-in kernel has) — flattens to the host-rows path (engine.MODEL_HOST: mcsas_hip_analyse_host_rows), and
    the row evaluation the library will call back (scatteringmodels.host_model_calc: set values, calcIntensity, restore) gives
    the reference's own model.calc() bit for bit and leaves the model's parameter values where they were."""
    _import_reference()
    import numpy
    from mcsas.bases.algorithm import RandomExponential
    from mcsas.utils.parameter import FitParameter, Parameter
    from mcsas.bases.model import SASModel
    from mcsas.dataobj.sasdata import SASData
    from mcsas_amd import engine

    class GuinierSphere(SASModel):
        shortName = "Guinier sphere (user model)"
        parameters = (FitParameter("rg", 5e-9, displayName="radius of gyration", generator=RandomExponential,
                                   valueRange=(1e-10, 1e-6), activeRange=(1e-9, 1e-7)),
                      Parameter("sld", 1e14, displayName="contrast", valueRange=(0., numpy.inf)))

        def __init__(self):
            super(GuinierSphere, self).__init__()
            self.rg.setActive(True)

        def volume(self):
            r = self.rg() * numpy.sqrt(5. / 3.)
            return (numpy.pi * 4. / 3.) * r**3

        def absVolume(self):
            return self.volume() * self.sld()**2

        def formfactor(self, dataset):
            q = self.getQ(dataset)
            return numpy.exp(-(q * self.rg())**2 / 6.)

    GuinierSphere.factory()
    m = GuinierSphere()
    m.rg.setActiveRange((2e-9, 5e-8))
    assert SM.is_host_model(m)
    setup = SM.setup_from_model(m)
    assert setup.model_id == engine.MODEL_HOST and setup.active_index == (0,) and setup.gen_kind == (1,)
    np.testing.assert_array_equal(setup.gen_lo, [2e-9]); np.testing.assert_array_equal(setup.gen_hi, [5e-8])
    q_nm = np.logspace(-2, 0.3, 40)
    d = SASData(title="syn", rawArray=np.stack([q_nm, np.ones(40), 0.01 * np.ones(40)], axis=1))
    d.config.nBin.setValue(0); d._reBin()
    pset = np.array([[3e-9], [1e-8], [4.9e-8], [1e-12], [1.0]])       # the last two are clipped by setValue (valueRange)
    before = m.rg()
    ref = m.calc(d, pset, 0.6666666)                                   # the reference's own ScatteringModel.calc
    cum, v, w, s_, rows = SM.host_model_calc(m, d, pset, 0.6666666, want_rows=True)
    assert m.rg() == before
    np.testing.assert_array_equal(cum, ref.cumInt); np.testing.assert_array_equal(v, ref.vset); np.testing.assert_array_equal(w, ref.wset)
    assert rows.shape == (5, 40) and np.isfinite(rows).all()
    np.testing.assert_array_equal(rows[3], SM.host_model_calc(m, d, [[1e-10]], 0.6666666, want_rows=True)[4][0])   # clipped to the range's edge

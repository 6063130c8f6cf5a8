"""Shared by the test modules: fixture loading and building the SAME model configuration for the
product (mcsas_amd) and for the checker (oracle)."""
import os
import numpy as np

import mcsas_amd
from mcsas_amd import engine
from oracle import mcsas_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


CASES = {
    "sphere": dict(cls=mcsas_amd.Sphere, omodel="sphere", active=["radius"]),
    "cyl_aspect": dict(cls=mcsas_amd.CylindersIsotropic, omodel="cyl", active=["radius", "aspect"]),
    "cyl_length": dict(cls=mcsas_amd.CylindersIsotropic, omodel="cyl", active=["radius", "length"],
                       fixed=dict(useAspect=False)),
    "ellcs": dict(cls=mcsas_amd.EllipsoidalCoreShell, omodel="ellcs", active=["a", "b", "t"]),
    "kholodenko": dict(cls=mcsas_amd.Kholodenko, omodel="kholodenko", active=["radius", "lenKuhn", "lenContour"]),
    "elliso": dict(cls=mcsas_amd.EllipsoidsIsotropic, omodel="elliso", active=["a", "aspect"]),
    "sphcs": dict(cls=mcsas_amd.SphericalCoreShell, omodel="sphcs", active=["radius", "t"]),
    "gausschain": dict(cls=mcsas_amd.GaussianChain, omodel="gausschain", active=["rg", "bp"]),
    "lmasphere": dict(cls=mcsas_amd.LMADenseSphere, omodel="lmasphere", active=["radius", "volFrac"]),
    # (no built-in kernel: the product runs it as a run-time plug-in, rows with an integral)
    "cylradiso": dict(cls=mcsas_amd.CylindersRadiallyIsotropic, omodel="cylradiso", active=["radius", "psiAngle"]),
}
GEN_CLS = {0: mcsas_amd.RandomUniform, 1: mcsas_amd.RandomExponential, 2: mcsas_amd.RandomExponential2,
           3: mcsas_amd.RandomExponential3}


def make_models(tag, lo=None, hi=None, gen=None, **fixed):
    """-> (product model instance, oracle ModelSpec) configured identically."""
    c = CASES[tag]
    fx = dict(c.get("fixed", {})); fx.update(fixed)
    m = c["cls"]()
    for p in m.params():
        if hasattr(p, "setActive"):
            p.setActive(p.name() in c["active"])
    for k, v in fx.items():
        getattr(m, k).setValue(v)
    n = len(c["active"])
    for i, name in enumerate(c["active"]):
        p = getattr(m, name)
        if lo is not None:
            p.setActiveRange((float(lo[i]), float(hi[i])))
        if gen is not None:
            p.setGenerator(GEN_CLS[int(gen[i])])
    ofx = {k: float(v) for k, v in fx.items()}
    spec = O.ModelSpec.make(c["omodel"], c["active"], lo if lo is not None else [0.0] * n,
                            hi if hi is not None else [np.inf] * n, gen, **ofx)
    return m, spec


def traj_setup(name):
    g = load(name)
    model = str(g["spec_model"])
    extra = {}
    if model in ("sphere", "cyl_aspect"):
        extra["sld"] = float(g["spec_sld"])
    if model == "cyl_aspect":
        extra["intDiv"] = float(g["spec_int_div"])
    if model == "ellcs":
        extra.update(eta_c=float(g["spec_eta_c"]), eta_s=float(g["spec_eta_s"]),
                     eta_sol=float(g["spec_eta_sol"]), intDiv=float(g["spec_int_div"]))
    if model == "elliso":
        extra.update(sld=float(g["spec_sld"]), intDiv=float(g["spec_int_div"]))
    if model == "sphcs":
        extra.update(eta_c=float(g["spec_eta_c"]), eta_s=float(g["spec_eta_s"]), eta_sol=float(g["spec_eta_sol"]))
    if model == "gausschain":
        extra.update(etas=float(g["spec_etas"]), k=float(g["spec_k"]))
    if model == "lmasphere":
        extra.update(sld=float(g["spec_sld"]), mf=float(g["spec_mf"]))
    if model == "cylradiso":
        extra.update(aspect=float(g["spec_aspect"]), sld=float(g["spec_sld"]), psiAngleDivisions=float(g["spec_divisions"]))
    m, spec = make_models(model, g["spec_lo"], g["spec_hi"], [int(x) for x in g["spec_gen"]], **extra)
    flags = dict(find_bg=bool(int(g["spec_find_bg"])) if "spec_find_bg" in g else True,
                 pos_bg=bool(int(g["spec_pos_bg"])) if "spec_pos_bg" in g else False,
                 start_from_min=bool(int(g["spec_from_min"])) if "spec_from_min" in g else False)
    ost = O.Settings(n_contrib=int(g["spec_n_contrib"]), n_reps=1, max_iter=int(g["spec_max_iter"]),
                     comp_exp=float(g["spec_comp_exp"]), conv_crit=float(g["spec_conv_crit"]), **flags)
    st = engine.Settings(n_contrib=ost.n_contrib, n_reps=1, max_iter=ost.max_iter, comp_exp=ost.comp_exp,
                         conv_crit=ost.conv_crit, find_background=flags["find_bg"],
                         positive_background=flags["pos_bg"], start_from_minimum=flags["start_from_min"],
                         max_retries=0)
    return g, m, spec, st, ost


class FakeData(object):
    """x0.limit holder for setup_from_model's startFromMinimum rule."""
    def __init__(self, q):
        class V: pass
        self.x0 = V(); self.x0.limit = [float(np.min(q)), float(np.max(q))]


SMEAR_CASES = (("trapz_slit", "trapezoid", False), ("trapz_pinhole", "trapezoid", True), ("gauss_slit", "gaussian", False))


def oracle_smearing(kind, two_d, n_steps, q, **widths):
    sm = O.Smearing(kind=kind, do_smear=True, n_steps=n_steps, two_d_coll=two_d, **widths)
    sm.prepare(q)
    return sm


def product_smearing(kind, two_d, n_steps, q, I=None, sigma=None, **widths):
    """-> (SASData with the smearing configured through the mirrored config API, SmearArgs)."""
    cfg = mcsas_amd.SASConfig(mcsas_amd.GaussianSmearing() if kind == "gaussian" else mcsas_amd.TrapezoidSmearing())
    d = mcsas_amd.SASData(q, np.ones_like(q) if I is None else I, np.ones_like(q) if sigma is None else sigma, config=cfg)
    sm = d.config.smearing
    sm.doSmear.setValue(True); sm.twoDColl.setValue(two_d); sm.nSteps.setValue(n_steps)
    if kind == "gaussian":
        sm.variance.setValue(widths["variance"])
    else:
        sm.penumbra.setValue(widths["penumbra"]); sm.umbra.setValue(widths["umbra"]); sm.penumbra.setValue(widths["penumbra"])
    d.updateConfig()
    return d, d.smearArgs(mcsas_amd.Sphere())


def traj_smearing(g):
    """Smearing of a trajectory fixture (g7_*): (oracle Smearing, product SmearArgs) or (None, None)."""
    if "smear_kind" not in g:
        return None, None
    kind, two_d, n = str(g["smear_kind"]), bool(int(g["smear_two_d"])), int(g["smear_n_steps"])
    widths = {k: float(g["smear_" + k]) for k in ("umbra", "penumbra", "variance") if "smear_" + k in g}
    osm = oracle_smearing(kind, two_d, n, g["data_q"], **widths)
    _, psm = product_smearing(kind, two_d, n, g["data_q"], **widths)
    return osm, psm


# ---- run-time model plug-ins (include/mcsas_hip.h: mcsas_hip_plugin_compile).  Two of the built-in models written again as
# plug-in text, operation for operation (csrc/models.h: Contrib<MCSAS_MODEL_GAUSS_CHAIN>, Contrib<MCSAS_MODEL_SPH_CS>), so
# that a plug-in chain can be compared bit for bit with its built-in twin and replayed against the reference's fixtures.
PLUGIN_SOURCES = {
    "gausschain": r"""
// models/gaussianchain.py:54-66; p = (rg, bp, etas, k)
__device__ double mcsas_plugin_volume(const double *p) { return p[3] * (p[0] * p[0]); }
__device__ double mcsas_plugin_absvolume(const double *p) { return mcsas_plugin_volume(p); }
__device__ double mcsas_plugin_surface(const double *p) { return 0.; }
__device__ double mcsas_plugin_formfactor(double q, const double *p) {
    const double beta = p[1] - mcsas_plugin_volume(p) * p[2];
    const double x = q * p[0], u = x * x;
    double f = sqrt(2.) * sqrt(expm1(-u) + u) / u;
    f *= beta;
    if (q <= 0.0) f = beta;
    return f;
}
""",
    "sphcs": r"""
// models/sphericalcoreshell.py:50-77; p = (radius, t, eta_c, eta_s, eta_sol)
__device__ double mcsas_plugin_volume(const double *p) { const double rt = p[0] + p[1]; return 4. / 3 * mcsas::PI * (rt * rt * rt); }
__device__ double mcsas_plugin_absvolume(const double *p) { return mcsas_plugin_volume(p); }
__device__ double mcsas_plugin_surface(const double *p) { return 0.; }
__device__ double mcsas_plugin_formfactor(double q, const double *p) {
    const double r = p[0], rt = p[0] + p[1];
    const double vr = (4. / 3 * mcsas::PI * (r * r * r)) / (4. / 3 * mcsas::PI * (rt * rt * rt));
    const double ds = p[3] - p[4], dc = p[3] - p[2];
    double sn, cs;
    const double xs = q * rt;
    mcsas::sincos_fast(xs, &sn, &cs);
    const double ks = mcsas::div_fast(ds * 3. * (sn - xs * cs), xs * xs * xs);
    const double xc = q * r;
    mcsas::sincos_fast(xc, &sn, &cs);
    const double kc = mcsas::div_fast(dc * 3. * (sn - xc * cs), xc * xc * xc);
    return ks - vr * kc;
}
""",
}


PLUGIN_SOURCES["elliso"] = r"""
// models/ellipsoidsisotropic.py:51-81, written the plain way: the orientation average as a loop inside the form factor;
// p = (a, useAspect, c, aspect, intDiv, sld).  A row of this model costs an integral per q point:
#define MCSAS_PLUGIN_ROW_CLASS 1
__device__ double mcsas_plugin_volume(const double *p) { const double rc = p[1] != 0. ? p[0] * p[3] : p[2]; return 4. / 3. * mcsas::PI * (p[0] * p[0]) * rc; }
__device__ double mcsas_plugin_absvolume(const double *p) { return mcsas_plugin_volume(p) * (p[5] * p[5]); }
__device__ double mcsas_plugin_surface(const double *p) { return 0.; }
__device__ double mcsas_plugin_formfactor(double q, const double *p) {
    const double ra = p[0], rc = p[1] != 0. ? p[0] * p[3] : p[2];
    const int K = (int)p[4];
    double sum = 0.;
    for (int k = 0; k < K; ++k) {
        const double al = (mcsas::PI / 2.) * (double)k / (double)(K - 1);       // np.linspace(0, pi/2, intDiv)
        double sa, ca, sx, cx;
        sincos(al, &sa, &ca);
        const double x = q * sqrt(ra * ra * (sa * sa) + rc * rc * (ca * ca));
        sincos(x, &sx, &cx);
        const double f = 3. * (sx - x * cx) / (x * x * x);
        sum += f * f * sa;
    }
    return sqrt(sum / (double)K);
}
"""
PLUGIN_SOURCES["sphere"] = r"""
// models/sphere.py:37-63; p = (radius, sld); canSmear = True
#define MCSAS_PLUGIN_CAN_SMEAR 1
__device__ double mcsas_plugin_volume(const double *p) { return (mcsas::PI * 4. / 3.) * (p[0] * p[0] * p[0]); }
__device__ double mcsas_plugin_absvolume(const double *p) { return mcsas_plugin_volume(p) * (p[1] * p[1]); }
__device__ double mcsas_plugin_surface(const double *p) { return 4. * mcsas::PI * p[0] * p[0]; }
__device__ double mcsas_plugin_formfactor(double q, const double *p) {
    const double x = q * p[0];
    double sn, cs;
    mcsas::sincos_fast(x, &sn, &cs);
    return 3. * (sn - x * cs) / (x * x * x);
}
"""


def plugin_twin(model, tag):
    """The same configured model instance, but as a user's own class: no built-in kernel id, its form factor as HIP text."""
    model.__class__ = type(type(model).__name__ + "AsPlugin", (type(model),), {"model_id": None, "hipSource": PLUGIN_SOURCES[tag]})
    return model

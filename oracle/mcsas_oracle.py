"""CPU restatement (numpy / scipy) of the McSAS Monte-Carlo hot path.

TEST INFRASTRUCTURE.  This module is the *checker*: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  The product (`mcsas_amd/`) never does; it fails
loudly when the HIP library is missing instead of falling back to anything in here.

Parity status: PINNED.  Every function below is checked in `tests/test_oracle_golden.py` against
fixtures under `tests/golden/` that were produced by running the real reference in the build
container (`oracle/make_golden.py`), and against the SASfit known-answer files the reference's own
(disabled) tests name (`sphere.py:68-75`, `kholodenko.py:98-102`).

Each function cites the reference file:line it restates (paths relative to
/root/reference/src/mcsas/).  Nothing here is copied; it is the same arithmetic written as flat
functions over plain arrays instead of the reference's Parameter/Algorithm object graph.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field

import numpy as np

# the reference calls numpy.trapz (cylindersisotropic.py:90); numpy >= 2 names it trapezoid
_trapz = getattr(np, "trapezoid", None) or np.trapz

# ----------------------------------------------------------------------------- model registry
SPHERE, CYL_ISO, ELL_CS, KHOLODENKO, ELL_ISO, SPH_CS, GAUSS_CHAIN, LMA_SPHERE = 0, 1, 2, 3, 4, 5, 6, 7
CYL_RAD_ISO = 8            # (oracle-side id only: the product runs this model as a run-time plug-in)
MODEL_IDS = {"sphere": SPHERE, "cyl": CYL_ISO, "ellcs": ELL_CS, "kholodenko": KHOLODENKO,
             "elliso": ELL_ISO, "sphcs": SPH_CS, "gausschain": GAUSS_CHAIN, "lmasphere": LMA_SPHERE,
             "cylradiso": CYL_RAD_ISO}

GEN_UNIFORM, GEN_EXP1, GEN_EXP2, GEN_EXP3 = 0, 1, 2, 3

# full parameter vectors, in the order of the reference's `parameters` tuples
PARAM_NAMES = {
    SPHERE: ("radius", "sld"),                                         # models/sphere.py:16-26
    CYL_ISO: ("radius", "useAspect", "length", "aspect", "intDiv", "sld"),  # cylindersisotropic.py:21-43
    ELL_CS: ("a", "b", "t", "eta_c", "eta_s", "eta_sol", "intDiv"),    # ellipsoidalcoreshell.py:19-52
    KHOLODENKO: ("radius", "lenKuhn", "lenContour"),                   # kholodenko.py:57-73
    ELL_ISO: ("a", "useAspect", "c", "aspect", "intDiv", "sld"),       # ellipsoidsisotropic.py:24-46
    SPH_CS: ("radius", "t", "eta_c", "eta_s", "eta_sol"),              # sphericalcoreshell.py:22-43
    GAUSS_CHAIN: ("rg", "bp", "etas", "k"),                            # gaussianchain.py:27-48
    LMA_SPHERE: ("radius", "volFrac", "mf", "sld"),                    # lmadensesphere.py:26-55
    CYL_RAD_ISO: ("radius", "aspect", "psiAngle", "psiAngleDivisions", "sld"),   # cylindersradiallyisotropic.py:20-41
}
PARAM_DEFAULTS = {
    SPHERE: (10e-9, 1e-6 * 1e20),
    CYL_ISO: (1e-9, 1.0, 10e-9, 10.0, 100.0, 1e-6 * 1e20),
    ELL_CS: (1e-9, 10e-9, 1e-9, 3.15e-6 * 1e20, 2.53e-6 * 1e20, 0.0, 100.0),
    KHOLODENKO: (1e-9, 1e-9, 2e-9),
    ELL_ISO: (1e-9, 1.0, 10e-9, 10.0, 100.0, 1e-6 * 1e20),
    SPH_CS: (1e-9, 1e-9, 3.16e-6 * 1e20, 2.53e-6 * 1e20, 0.0),
    GAUSS_CHAIN: (1e-9, 100e-9, 1e-6 * 1e20, 1.0),
    LMA_SPHERE: (1e-9, 0.10, -1.0, 1e-6 * 1e20),
    CYL_RAD_ISO: (1e-9, 10.0, 0.17, 303.0, 1e-6 * 1e20),
}
# valueRange (clip range applied by Parameter.setValue, bases/algorithm/parameter.py:405-414,489-495)
PARAM_VALUE_RANGE = {
    SPHERE: ((0.0, np.inf), (0.0, np.inf)),
    CYL_ISO: ((0.1e-9, np.inf), (0.0, 1.0), (0.1e-9, 1e10 * 1e-9), (1e-3, 1e3), (1.0, 1e4), (0.0, np.inf)),
    ELL_CS: ((0.0, np.inf),) * 6 + ((0.0, 1e4),),
    KHOLODENKO: ((0.0, np.inf),) * 3,
    ELL_ISO: ((0.1e-9, 1e10 * 1e-9), (0.0, 1.0), (0.1e-9, 1e10 * 1e-9), (1e-3, 1e3), (0.0, 1e4), (0.0, 1e-2 * 1e20)),
    SPH_CS: ((0.0, np.inf),) * 5,
    GAUSS_CHAIN: ((0.0, np.inf),) * 4,
    LMA_SPHERE: ((0.0, np.inf), (0.001e-2, 1.0), (-1.0, 1e6), (0.0, np.inf)),
    CYL_RAD_ISO: ((0.1e-9, np.inf), (0.1, np.inf), (0.01, 2 * np.pi + 0.01), (1.0, np.inf), (0.0, np.inf)),
}
PARAM_DEFAULT_GEN = {
    SPHERE: {"radius": GEN_UNIFORM},
    CYL_ISO: {"radius": GEN_EXP1, "length": GEN_EXP1, "aspect": GEN_EXP1},
    ELL_CS: {"a": GEN_EXP1, "b": GEN_EXP1, "t": GEN_EXP1},
    KHOLODENKO: {"radius": GEN_EXP1, "lenKuhn": GEN_UNIFORM, "lenContour": GEN_UNIFORM},
    ELL_ISO: {"a": GEN_EXP1, "c": GEN_EXP1, "aspect": GEN_EXP1},
    SPH_CS: {"radius": GEN_EXP1, "t": GEN_EXP1},
    GAUSS_CHAIN: {"rg": GEN_EXP1, "bp": GEN_UNIFORM, "etas": GEN_UNIFORM, "k": GEN_UNIFORM},
    LMA_SPHERE: {"radius": GEN_UNIFORM, "volFrac": GEN_UNIFORM},
    CYL_RAD_ISO: {"radius": GEN_EXP1, "aspect": GEN_UNIFORM, "psiAngle": GEN_UNIFORM},
}


@dataclass
class ModelSpec:
    """Flat description of a configured ScatteringModel instance: which parameters are active
    (columns of `rset`), their generator ranges/kinds and the values of everything else."""
    model_id: int
    active: tuple            # indices into PARAM_NAMES[model_id], ascending (activeParams() order)
    lo: np.ndarray           # activeRange ∩ valueRange, lower (utils/parameter.py:715-728, parameter.py:66-84)
    hi: np.ndarray
    gen: tuple               # GEN_* per active parameter
    values: np.ndarray       # full parameter vector (inactive entries are used as-is)
    smear: object = None     # a prepared Smearing (data.config.smearing + data.locs) or None

    @property
    def n_active(self):
        return len(self.active)

    @staticmethod
    def make(model, active, lo, hi, gen=None, **fixed):
        mid = MODEL_IDS[model] if isinstance(model, str) else int(model)
        names = PARAM_NAMES[mid]
        idx = tuple(sorted(names.index(a) for a in active))
        order = [names[i] for i in idx]
        lo_d = dict(zip(active, lo)); hi_d = dict(zip(active, hi))
        gen_d = dict(zip(active, gen)) if gen is not None else {}
        vals = np.array(PARAM_DEFAULTS[mid], dtype=float)
        for k, v in fixed.items():
            vals[names.index(k)] = float(v)
        lo_a, hi_a, g_a = [], [], []
        for i, n in zip(idx, order):
            vr = PARAM_VALUE_RANGE[mid][i]
            lo_a.append(max(vr[0], lo_d[n])); hi_a.append(min(vr[1], hi_d[n]))
            g_a.append(gen_d.get(n, PARAM_DEFAULT_GEN[mid][n]))
        return ModelSpec(mid, idx, np.array(lo_a, float), np.array(hi_a, float), tuple(g_a), vals)


# ----------------------------------------------------------------------------- form factors
def ff_sphere(q, radius):
    """models/sphere.py:55-63."""
    qr = q * radius
    return 3. * (np.sin(qr) - qr * np.cos(qr)) / (qr**3.)


def ff_cylinders_isotropic(q, radius, half_length, int_div):
    """models/cylindersisotropic.py:50-90 (x end points replaced by analytic limits :79-82)."""
    from scipy.special import j1
    x, step = np.linspace(0., 1., int(int_div), endpoint=True, retstep=True)
    x[0] = 0.5
    x[-1] = 0.5
    qrs = np.outer(q, radius * np.sqrt(1. - x**2.))
    qlx = np.outer(q, 2. * half_length * x)
    fsplit = (j1(qrs) * np.sin(qlx / 2.)) / (qrs * qlx)
    fsplit[:, 0] = 0.5 * (j1(q * radius) / (q * radius))
    fsplit[:, -1] = np.sin(q * half_length) / (q * half_length)
    return np.sqrt(16 * _trapz(fsplit**2, dx=step, axis=1))


def ff_ellipsoidal_core_shell(q, a, b, t, eta_c, eta_s, eta_sol, int_div):
    """models/ellipsoidalcoreshell.py:59-90."""
    def j1(x):
        return (np.sin(x) - x * np.cos(x)) / (x**2)
    mu = np.linspace(0., 1., int(int_div))
    vc = 4. / 3. * np.pi * a * b**2.
    vt = 4. / 3. * np.pi * (a + t) * (b + t)**2.
    v_ratio = vc / vt
    xc = np.outer(q, np.sqrt(a**2 * mu**2 + b**2 * (1 - mu**2)))
    xt = np.outer(q, np.sqrt((a + t)**2 * mu**2 + (b + t)**2 * (1 - mu**2)))
    fsplit = ((eta_c - eta_s) * v_ratio * (3 * j1(xc) / xc)
              + (eta_s - eta_sol) * 1. * (3 * j1(xt) / xt))
    return np.sqrt(np.mean(fsplit**2, axis=1))


def _kho_core(z, q_value, kuhn, x):
    """models/kholodenko.py:16-30."""
    if z <= 0.0 or x <= 0.0:
        return 1.0
    ratio = 3.0 / kuhn
    if q_value < ratio:
        e = math.sqrt(1.0 - q_value * q_value * kuhn * kuhn / 9.)
        fz = math.sinh(e * z) / (e * math.sinh(z))
    elif q_value > ratio:
        f = math.sqrt(q_value * q_value * kuhn * kuhn / 9. - 1.0)
        fz = math.sin(f * z) / (f * math.sinh(z))
    else:
        fz = z / math.sinh(z)
    return fz * (2. / x) * (1.0 - z / x)


def ff_kholodenko(q, radius, len_kuhn, len_contour):
    """models/kholodenko.py:32-49,81-90: QUADPACK QAGS, epsrel 1e-10, limit 10000."""
    from scipy.integrate import quad
    from scipy.special import j1
    x = 3. * len_contour / len_kuhn
    out = np.empty(len(q))
    for i, qv in enumerate(q):
        res = quad(_kho_core, 0, x, args=(float(qv), len_kuhn, x),
                   limit=10000, full_output=1, epsabs=0.0, epsrel=1e-10)
        p0 = math.sqrt(res[0])
        u = qv * radius
        pcs = 1.0 if u <= 0.0 else 2. * j1(u) / u
        out[i] = p0 * pcs
    return out


def ff_ellipsoids_isotropic(q, ra, rc, int_div):
    """models/ellipsoidsisotropic.py:51-73."""
    al = np.linspace(0., np.pi / 2., int(int_div))
    qrp = np.outer(q, np.sqrt(ra**2 * np.sin(al)**2 + rc**2 * np.cos(al)**2))
    fsplit = 3. * (np.sin(qrp) - qrp * np.cos(qrp)) / (qrp**3.)
    return np.sqrt(np.mean(fsplit**2 * np.sin(al), axis=1))


def ff_cylinders_radially_isotropic(q, radius, aspect, psi_angle, divisions, psi_range):
    """models/cylindersradiallyisotropic.py:49-74: the in-plane orientation average over psi = linspace(valueRange of psiAngle)."""
    from scipy.special import j1
    psi = np.linspace(psi_range[0], psi_range[1], int(divisions))
    qrs = np.outer(q, radius * np.sin(psi - psi_angle))
    qlc = np.outer(q, radius * aspect * np.cos(psi - psi_angle))
    fsplit = 2. * j1(qrs) / qrs * np.sin(qlc) / qlc
    return np.sqrt(np.mean(fsplit**2, axis=1))


def ff_spherical_core_shell(q, r, t, eta_c, eta_s, eta_sol):
    """models/sphericalcoreshell.py:50-69."""
    def k(q, rr, d_eta):
        qr = np.outer(q, rr)
        return d_eta * 3 * (np.sin(qr) - qr * np.cos(qr)) / (qr)**3
    vc = 4. / 3 * np.pi * r**3
    vt = 4. / 3 * np.pi * (r + t)**3
    v_ratio = vc / vt
    ks = k(q, r + t, eta_s - eta_sol)
    kc = k(q, r, eta_s - eta_c)
    return (ks - v_ratio * kc).flatten()


def ff_gaussian_chain(q, rg, bp, etas, k):
    """models/gaussianchain.py:54-61."""
    beta = bp - (k * rg**2) * etas
    u = (q * rg)**2
    result = np.sqrt(2.) * np.sqrt(np.expm1(-u) + u) / u
    result *= beta
    result[q <= 0.0] = beta
    return result


def ff_lma_dense_sphere(q, r, mu, mf):
    """models/lmadensesphere.py:68-100."""
    if mf == -1:
        mf = (0.634 / mu)**(1. / 3)

    def sfg(A, mu):
        alpha = (1 + 2 * mu)**2 / (1 - mu)**4
        beta = -6 * mu * (1 + mu / 2)**2 / (1 - mu)**4
        gamma = mu * alpha / 2
        return (alpha * (np.sin(A) - A * np.cos(A)) / A**2
                + beta * (2 * A * np.sin(A) + (2 - A**2) * np.cos(A) - 2) / A**3
                + gamma * (-1 * A**4 * np.cos(A) + 4 * ((3 * A**2 - 6) * np.cos(A)
                                                       + (A**3 - 6 * A) * np.sin(A) + 6)) / A**5)
    qr = q * r
    result = 3. * (np.sin(qr) - qr * np.cos(qr)) / (qr**3.)
    rhsq = 2. * q * (mf * r)
    S = ((1. + 24. * mu * sfg(rhsq, mu) / rhsq))**(-1)
    return np.sqrt(result**2 * S)


# ----------------------------------------------------------------------------- beam-profile smearing
CAN_SMEAR = (SPHERE, LMA_SPHERE)          # canSmear = True: models/sphere.py:15, models/lmadensesphere.py:23


class Smearing:
    """SmearingConfig + TrapezoidSmearing / GaussianSmearing (dataobj/sasconfig.py:17-260) and
    SASConfig.prepareSmearing (:308-339): integration offsets qOffset[K], profile weights[K] and the
    evaluation points locs[Q][K] of the smeared intensity (SASData.locs, dataobj/sasdata.py:165).

    Quirks kept: the pinhole offsets use `ceil(n/2)` points per side (the reference passes that float
    to numpy.logspace, which numpy < 1.18 truncated to int and numpy 2 rejects); the Gaussian profile
    uses the parameter called `variance` as the standard deviation (scipy.stats.norm.pdf(scale=...));
    parity of this path is pinned by tests/golden/smearing.npz, generated from the reference with
    numpy.logspace wrapped to take that float (oracle/make_golden.py)."""

    def __init__(self, kind="trapezoid", do_smear=False, n_steps=25, two_d_coll=False,
                 umbra=0., penumbra=0., variance=0.):
        self.kind, self.do_smear, self.n_steps, self.two_d_coll = kind, bool(do_smear), int(n_steps), bool(two_d_coll)
        self.umbra, self.penumbra, self.variance = float(umbra), float(penumbra), float(variance)
        self.q_offset = self.weights = self.locs = None

    def input_valid(self):                                   # sasconfig.py:94-96 / :198-200
        if self.kind == "trapezoid":
            return self.umbra > 0. and self.penumbra > self.umbra
        return self.variance > 0.

    @property
    def active(self):                                        # the test in sasmodel.py:56-60
        return self.do_smear and self.input_valid() and self.q_offset is not None

    @staticmethod
    def half_trapz_pdf(x, c, d):                             # sasconfig.py:104-120
        x = np.abs(x)
        pdf = x * 0.
        pdf[x < c] = 1.
        if d > c:
            m = (c <= x) & (x < d)
            pdf[m] = (1. / (d - c)) * (d - x[m])
        return pdf * (1. / (d + c))

    def set_int_points(self, q):                             # sasconfig.py:122-149 / :209-233
        n = self.n_steps
        if self.kind == "trapezoid":
            lo, hi = np.log10(q.min() / 5.), np.log10(self.penumbra / 2.)
        else:
            lo, hi = np.log10(q.min() / 3.), np.log10(2.5 * self.variance)
        if self.two_d_coll:
            off = np.logspace(lo, hi, num=int(np.ceil(n / 2.)))
            off = np.concatenate((-off[::-1], [0, ], off))
        else:
            off = np.concatenate(([0, ], np.logspace(lo, hi, num=n)))
        if self.kind == "trapezoid":
            y = self.half_trapz_pdf(off, self.umbra, self.penumbra)
        else:
            y = np.exp(-0.5 * (off / self.variance) ** 2) / (self.variance * np.sqrt(2. * np.pi))
        self.q_offset, self.weights = off, y

    def prepare(self, q):                                    # SASConfig.prepareSmearing, sasconfig.py:308-339
        q = np.asarray(q, dtype=float)
        self.q_offset = self.weights = None
        if not (self.input_valid() and self.do_smear):
            self.locs = q
            return q
        self.set_int_points(q)
        if not self.two_d_coll:
            self.locs = np.sqrt(np.add.outer(q ** 2, self.q_offset ** 2))
        else:
            self.locs = np.add.outer(q, self.q_offset)
        return self.locs


def _trapz_x(y, x):
    """numpy.trapz(y, x=x, axis=1)."""
    d = np.diff(x)
    return ((y[:, 1:] + y[:, :-1]) * d / 2.0).sum(axis=1)


def _clip_full(spec: ModelSpec, row):
    """Full parameter vector for one contribution: active columns set from `row`, each clipped
    into its valueRange as Parameter.setValue does (bases/algorithm/parameter.py:405-414)."""
    p = spec.values.copy()
    for col, i in enumerate(spec.active):
        lo, hi = PARAM_VALUE_RANGE[spec.model_id][i]
        p[i] = min(max(row[col], lo), hi)
    return p


def calc_intensity(spec: ModelSpec, q, row, comp_exp):
    """SASModel.calcIntensity (bases/model/sasmodel.py:46-79) for one contribution: returns
    (it[Q], v, w, s) with it = F² · volume()^(2c), v = absVolume(); with an active smearing
    configuration and a canSmear model, F is evaluated at data.locs[Q][K] and
    it = 2 · trapz(F² · w · weights, x = qOffset) (:56-73)."""
    p = _clip_full(spec, row)
    mid = spec.model_id
    sm = spec.smear if (spec.smear is not None and spec.smear.active and mid in CAN_SMEAR) else None
    if sm is not None:
        assert sm.locs.shape[0] == len(q)
        q = sm.locs
    if mid == SPHERE:
        r, sld = p
        vol = (np.pi * 4. / 3.) * r**3                         # sphere.py:39-46
        v = vol * sld**2                                       # sphere.py:47-53
        s = 4. * np.pi * r * r                                 # sphere.py:32-37
        ff = ff_sphere(q, r)
    elif mid == CYL_ISO:
        r, use_aspect, length, aspect, int_div, sld = p
        hl = r * aspect if use_aspect else 0.5 * length        # cylindersisotropic.py:65-68
        vol = np.pi * r**2 * (hl * 2.)                         # :92-98
        v = vol * sld**2                                       # :100-101
        s = 0
        ff = ff_cylinders_isotropic(q, r, hl, int_div)
    elif mid == ELL_CS:
        a, b, t, eta_c, eta_s, eta_sol, int_div = p
        vol = 4. / 3 * np.pi * (a + t) * (b + t)**2            # ellipsoidalcoreshell.py:92-94
        v = vol                                                # :96-97
        s = 0
        ff = ff_ellipsoidal_core_shell(q, a, b, t, eta_c, eta_s, eta_sol, int_div)
    elif mid == KHOLODENKO:
        r, lk, lc = p
        vol = np.pi * lc * r**2                                # kholodenko.py:92-94
        v = vol
        s = 0
        ff = ff_kholodenko(q, r, lk, lc)
    elif mid == ELL_ISO:
        ra, use_aspect, c, aspect, int_div, sld = p
        rc = ra * aspect if use_aspect else c                  # ellipsoidsisotropic.py:63-66
        vol = 4. / 3. * np.pi * ra**2. * rc                    # :75-81
        v = vol * sld**2
        s = 0
        ff = ff_ellipsoids_isotropic(q, ra, rc, int_div)
    elif mid == CYL_RAD_ISO:
        r, aspect, psi_a, div, sld = p
        vol = np.pi * r**2 * (2. * r * aspect)                 # cylindersradiallyisotropic.py:76-78
        v = vol * sld**2                                       # :80-81
        s = 0
        ff = ff_cylinders_radially_isotropic(q, r, aspect, psi_a, div, PARAM_VALUE_RANGE[CYL_RAD_ISO][2])
    elif mid == SPH_CS:
        r, t, eta_c, eta_s, eta_sol = p
        vol = 4. / 3 * np.pi * (r + t)**3                      # sphericalcoreshell.py:71-73
        v = vol
        s = 0
        ff = ff_spherical_core_shell(q, r, t, eta_c, eta_s, eta_sol)
    elif mid == GAUSS_CHAIN:
        rg, bp, etas, kk = p
        vol = kk * rg**2                                       # gaussianchain.py:63-65
        v = vol
        s = 0
        ff = ff_gaussian_chain(q, rg, bp, etas, kk)
    elif mid == LMA_SPHERE:
        r, mu, mf, sld = p
        vol = (np.pi * 4. / 3.) * r**3                         # lmadensesphere.py:61-63
        v = vol * sld**2
        s = 0
        ff = ff_lma_dense_sphere(q, r, mu, mf)
    else:
        raise ValueError("unknown model id %r" % mid)
    w = vol**(2 * comp_exp)                                    # sasmodel.py:37-44
    if sm is not None:
        return 2 * _trapz_x(ff**2 * w * sm.weights, sm.q_offset), v, w, s    # sasmodel.py:72-73
    return ff**2 * w, v, w, s


def model_calc(spec: ModelSpec, q, pset, comp_exp):
    """ScatteringModel.calc (bases/model/scatteringmodel.py:79-105): sequential accumulation."""
    pset = np.asarray(pset, dtype=float).reshape(-1, spec.n_active)
    cum = np.zeros(len(q))
    vset = np.zeros(len(pset)); wset = np.zeros(len(pset)); sset = np.zeros(len(pset))
    for i, row in enumerate(pset):
        it, vset[i], wset[i], sset[i] = calc_intensity(spec, q, row, comp_exp)
        cum += it
    return cum, vset, wset, sset


# ----------------------------------------------------------------------------- random numbers
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
_M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon et al., SC'11). Inputs: uint32-valued arrays/scalars."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0 = int(k0) & 0xFFFFFFFF; k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _M32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _M32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & _M32, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & _M32, lo0
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_uniform(seed, chain, idx):
    """The build's device RNG (no reference counterpart: the reference uses the unseeded global
    MT19937, numbergenerator.py:31).  Draw `idx` of chain `chain`: Philox counter =
    (idx>>1 lo, idx>>1 hi, chain, 0), key = seed lo/hi; even idx uses words 0,1, odd idx 2,3;
    53-bit double = ((a>>5)·2^26 + (b>>6)) / 2^53."""
    idx = np.asarray(idx, dtype=np.uint64)
    blk = idx >> np.uint64(1)
    r0, r1, r2, r3 = philox4x32_10(blk & _M32, blk >> np.uint64(32), np.uint64(chain), np.uint64(0),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    odd = (idx & np.uint64(1)).astype(bool)
    a = np.where(odd, r2, r0); b = np.where(odd, r3, r1)
    return ((a >> np.uint64(5)).astype(np.float64) * 67108864.0
            + (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


class ReplayStream:
    """Serves a pre-drawn uniform stream in order (what numpy.random.uniform would have returned)."""
    def __init__(self, values, pos=0):
        self.values = np.asarray(values, dtype=float); self.pos = pos

    def draw(self, count):
        out = self.values[self.pos:self.pos + count]
        if len(out) != count:
            raise IndexError("replay stream exhausted")
        self.pos += count
        return out.copy()


class PhiloxStream:
    def __init__(self, seed, chain, pos=0):
        self.seed, self.chain, self.pos = int(seed), int(chain), pos

    def draw(self, count):
        out = philox_uniform(self.seed, self.chain, np.arange(self.pos, self.pos + count, dtype=np.uint64))
        self.pos += count
        return out


def transform(gen_kind, u):
    """NumberGenerator.get (bases/algorithm/numbergenerator.py:28-31,168-191) applied to raw
    uniforms u in [0,1): RandomExponentialK maps u -> (10^(K·u) − 1) / 10^K."""
    if gen_kind == GEN_UNIFORM:
        return u
    upper = float(gen_kind)                   # 1, 2 or 3 decades; lower = 0
    rs = 10**(0. + (upper - 0.) * u)          # numpy.random.uniform(lower, upper) = lower + (upper-lower)·u
    return (rs - 1) / (10**(upper - 0.))


def generate_parameters(spec: ModelSpec, stream, count=1):
    """ScatteringModel.generateParameters (scatteringmodel.py:117-127) + generateValues
    (bases/algorithm/parameter.py:66-84): column by column, `count` draws per active parameter."""
    out = np.zeros((count, spec.n_active))
    for col in range(spec.n_active):
        vals = transform(spec.gen[col], stream.draw(count))
        out[:, col] = vals * (spec.hi[col] - spec.lo[col]) + spec.lo[col]
    return out


# ----------------------------------------------------------------------------- scale/background fit
def bgfit_closed(I, sigma, C, find_bg=True, pos_bg=False):
    """Closed-form minimiser of Σ((I − A·C − b)/σ)² (what leastsq converges to in
    backgroundscalingfit.py:94-103).  pos_bg: b is replaced by |b| in the residual (:59-63), so a
    negative free optimum collapses onto the b = 0 boundary."""
    w = 1.0 / (sigma * sigma)
    sw, swc, swcc = w.sum(), (w * C).sum(), (w * C * C).sum()
    swi, swic = (w * I).sum(), (w * I * C).sum()
    if find_bg:
        det = sw * swcc - swc * swc
        A = (sw * swic - swi * swc) / det
        b = (swi - A * swc) / sw
        if pos_bg and b < 0:
            A, b = swic / swcc, 0.0
    else:
        A, b = swic / swcc, 0.0
    return np.array([A, b])


def bgfit_calc(I, sigma, C, sc, find_bg=True, pos_bg=False, ver=2, num_params=1, method="leastsq"):
    """BackgroundScalingFit.calc (mcsas/backgroundscalingfit.py:112-139).
    method='leastsq' follows the reference call for call (scipy MINPACK / Nelder-Mead);
    method='closed' is the closed form the HIP kernels use.  Returns (sc, conval, aGoFs)."""
    err = np.array(sigma, dtype=float).flatten()
    err[err == 0.0] = 1.
    I = np.asarray(I, dtype=float).flatten()
    sc = np.array(sc, dtype=float)
    if not len(I):
        return sc, 1., 1.

    def scaled(xsc):                                         # dataScaled :86-92
        if find_bg:
            return C * xsc[0] + (abs(xsc[1]) if pos_bg else xsc[1])
        return C * xsc[0]

    def chisqr(calc):                                        # chiSqr :72-77 (python builtin sum)
        return sum(((I - calc) / err)**2) / len(I)

    if method == "closed":
        sc = bgfit_closed(I, err, C, find_bg, pos_bg)
    elif ver == 2:                                           # fitLM :94-103
        from scipy import optimize
        if not find_bg:
            func = lambda s: (I - s[0] * C) / err
        elif pos_bg:
            func = lambda s: (I - s[0] * C - abs(s[1])) / err
        else:
            func = lambda s: (I - s[0] * C - s[1]) / err
        sc, _ = optimize.leastsq(func, sc, full_output=False)
    else:                                                    # fitSimplex :105-110
        from scipy import optimize
        sc = optimize.fmin(lambda xsc: chisqr(scaled(xsc)), sc, full_output=False, disp=0)
    sc = np.array(sc, dtype=float)
    if not find_bg:
        sc[1] = 0.0
    elif pos_bg:
        sc[1] = abs(sc[1])
    fit = scaled(sc)
    conval = chisqr(fit)
    agofs = sum((I - fit)**2) / sum(err**2)                  # aGoFsAlpha :79-84
    agofs *= len(I) / (len(I) - num_params)
    return sc, conval, agofs


# ----------------------------------------------------------------------------- the MC chain
@dataclass
class Settings:
    """Algorithm settings (mcsas/mcsasparameters.json:2-103) that reach the hot path."""
    n_contrib: int = 300
    n_reps: int = 10
    max_iter: int = 100000
    comp_exp: float = 0.6666666
    conv_crit: float = 1.0
    find_bg: bool = True
    pos_bg: bool = False
    start_from_min: bool = False
    max_retries: int = 5
    show_incomplete: bool = False


@dataclass
class ChainResult:
    rset: np.ndarray
    fit: np.ndarray
    conval: float
    num_iter: int
    num_moves: int
    scaling: float
    background: float
    accepted: list = field(default_factory=list)
    elapsed: float = 0.0


def mc_fit(spec: ModelSpec, q, I, sigma, f_limit, x0_limit, st: Settings, stream,
           method="leastsq", cache=True, stop=None):
    """McSAS.mcFit (mcsas/mcsas.py:287-439): one chain.

    `cache=True` keeps each contribution's intensity vector instead of re-evaluating `old`
    (mcsas.py:362); the values are bit-identical because calc_intensity is deterministic, it only
    saves time.  `f_limit` = data.f.limit, `x0_limit` = data.x0.limit (datavector.py:46-54)."""
    N = st.n_contrib
    t0 = time.time()
    if st.start_from_min:                                               # mcsas.py:310-315
        rset = np.zeros((N, spec.n_active))
        for col in range(spec.n_active):
            mb = min(spec.lo[col], spec.hi[col])
            if mb == 0:
                mb = np.pi / x0_limit[1]
            rset[:, col] = np.ones(N) * mb * .5
    else:
        rset = generate_parameters(spec, stream, N)                     # :317
    rows = np.empty((N, len(q)))
    ft = np.zeros(len(q))
    wset = np.zeros(N)
    for i in range(N):                                                  # :319 (model.calc)
        rows[i], _, wset[i], _ = calc_intensity(spec, q, rset[i], st.comp_exp)
        ft += rows[i]
    sc = np.array((1.0, f_limit[0]))                                    # :327-330
    if len(ft) and ft.max() != 0.0:
        sc[0] = f_limit[1] / ft.max()
    kw = dict(find_bg=st.find_bg, pos_bg=st.pos_bg, num_params=spec.n_active, method=method)
    sc, conval, _ = bgfit_calc(I, sigma, ft, sc, ver=1, **kw)           # :339
    sc, conval, _ = bgfit_calc(I, sigma, ft, sc, **kw)                  # :343
    num_moves = num_iter = 0
    ri = 0
    accepted = []
    while (N > 1 and conval > st.conv_crit and num_iter < st.max_iter
           and not (stop is not None and stop())):                      # :354-357
        rt = generate_parameters(spec, stream, 1)                       # :358
        new, _, wnew, _ = calc_intensity(spec, q, rt[0], st.comp_exp)   # :360
        old = rows[ri] if cache else calc_intensity(spec, q, rset[ri], st.comp_exp)[0]   # :362
        test = ft - old + new                                           # :367
        sct, convalt, _ = bgfit_calc(I, sigma, test, sc, **kw)          # :376
        if convalt < conval:                                            # :379-390
            rset[ri], sc, conval = rt[0], sct, convalt
            ft, wset[ri] = test, wnew
            rows[ri] = new
            accepted.append(num_iter)
            num_moves += 1
        ri = (ri + 1) % N                                               # :403-404
        num_iter += 1
    sc, conval, _ = bgfit_calc(I, sigma, ft, sc, **kw)                  # :424-425
    fit = ft * sc[0] + sc[1]                                            # :430
    return ChainResult(rset, fit, float(conval), num_iter, num_moves, float(sc[0]), float(sc[1]),
                       accepted, time.time() - t0 + 1e-3)


def analyse(spec: ModelSpec, q, I, sigma, f_limit, x0_limit, st: Settings, streams, method="leastsq"):
    """McSAS.analyse (mcsas/mcsas.py:191-285).  `streams`: one stream shared by all reps (the
    reference's single global RNG) or a list with one stream per rep (the build's layout).
    Returns (result dict | None, per-rep info)."""
    R, N = st.n_reps, st.n_contrib
    contribs = np.zeros((N, spec.n_active, R))
    num_iter = np.zeros(R); scalings = np.zeros(R); backgrounds = np.zeros(R); times = np.zeros(R)
    meas = np.zeros([1, len(q), R])
    info = []
    for nr in range(R):
        stream = streams[nr] if isinstance(streams, (list, tuple)) else streams
        start_pos = getattr(stream, "pos", 0)
        t0 = time.time()
        nt = 0
        convergence = np.inf
        res = None
        while convergence > st.conv_crit:                               # :220
            if nt > st.max_retries:                                     # :221-230
                if st.show_incomplete:
                    break
                return None, info
            res = mc_fit(spec, q, I, sigma, f_limit, x0_limit, st, stream, method=method)
            contribs[:, :, nr], meas[0, :, nr], convergence = res.rset, res.fit, res.conval
            nt += 1
        num_iter[nr], scalings[nr], backgrounds[nr] = res.num_iter, res.scaling, res.background
        times[nr] = time.time() - t0
        info.append(dict(start=start_pos, end=getattr(stream, "pos", 0), attempts=nt,
                         conval=res.conval, num_moves=res.num_moves))
    ddof = 1 if R > 1 else 0                                            # :265-267
    result = dict(contribs=contribs,
                  fitMeasValMean=meas.mean(axis=2), fitMeasValStd=meas.std(axis=2),
                  fitX0=q, dataX0=q, dataMean=I, dataStd=sigma,
                  scaling=(scalings.mean(), scalings.std(ddof=ddof)),
                  background=(backgrounds.mean(), backgrounds.std(ddof=ddof)),
                  times=times, numIter=num_iter.mean())
    return result, info


# ----------------------------------------------------------------------------- post-fit histogram
YWEIGHTS = ("vol", "num", "int", "surf")


def fractions(spec: ModelSpec, q, I, sigma, f_limit, st: Settings, contribs, method="leastsq"):
    """McSAS.histogram, first half (mcsas/mcsas.py:519-609): per-rep volume/number/intensity/surface
    fractions and the per-contribution minimum-visibility limits."""
    N, _, R = contribs.shape
    vf = np.zeros((N, R)); nf = np.zeros((N, R)); qf = np.zeros((N, R)); sf = np.zeros((N, R))
    mv = np.zeros((N, R)); mn = np.zeros((N, R)); mq = np.zeros((N, R)); ms = np.zeros((N, R))
    scaling = np.zeros((2, R))
    err = np.array(sigma, dtype=float)
    for ri in range(R):
        rset = contribs[:, :, ri]
        cum, vset, wset, sset = model_calc(spec, q, rset, st.comp_exp)              # :552
        sc = np.array([f_limit[1] / cum.max(), f_limit[0]])                         # :557
        sc, conval, _ = bgfit_calc(I, sigma, cum, sc, find_bg=st.find_bg, pos_bg=st.pos_bg,
                                   num_params=spec.n_active, method=method)         # :559
        scaling[:, ri] = sc
        vf[:, ri] = wset * sc[0] / vset                                             # :565, modeldata.py:57-61
        tv = sum(vf[:, ri])
        nf[:, ri] = vf[:, ri] / vset
        tn = sum(nf[:, ri])
        qf[:, ri] = vf[:, ri] * vset
        tq = sum(qf[:, ri])
        sf[:, ri] = nf[:, ri] * sset
        ts = sum(sf[:, ri])
        for c in range(N):                                                          # :575-594
            part = calc_intensity(spec, q, rset[c], st.comp_exp)[0]
            weighted = err * vf[c, ri]
            scaled = sc[0] * part
            ind = scaled != 0.
            mv[c, ri] = (weighted[ind] / scaled[ind]).min()
            mn[c, ri] = mv[c, ri] / vset[c]
            mq[c, ri] = mn[c, ri] * mv[c, ri] * mv[c, ri]
            ms[c, ri] = mn[c, ri] * sset[c]
        if 0 != tn:                                                                 # :596-604
            nf[:, ri] /= tn; mn[:, ri] /= tn
        if 0 != tq:
            qf[:, ri] /= tq; mq[:, ri] /= tq
        if 0 != ts:
            sf[:, ri] /= ts; ms[:, ri] /= ts
    return dict(vol=(vf, mv), num=(nf, mn), int=(qf, mq), surf=(sf, ms)), scaling


def histogram_calc(contribs, param_index, frac, lower, upper, bin_count, xscale="log", yweight="vol"):
    """Histogram.calc … _calcCDF and Moments (utils/parameter.py:20-154,349-479)."""
    N, _, R = contribs.shape
    if "lin" in xscale:                                                   # _setXLowerEdge :349-362
        edges = np.linspace(lower, upper, bin_count + 1)
    else:
        edges = np.logspace(np.log10(lower), np.log10(upper), bin_count + 1)
    fr, min_req = frac[yweight]
    bins_l, obs_l, cdf_l = [], [], []
    for ri in range(R):                                                   # _calcRepetitions :424-439
        par = contribs[:, param_index, ri]
        bins = np.zeros(bin_count); bobs = np.zeros(bin_count)
        for bi in range(bin_count):                                       # _calcBins/_calcBin :441-469
            mask = (par >= edges[bi]) * (par < edges[bi + 1])
            val = sum(fr[mask, ri])
            if np.isnan(val):
                val = 0.
            bins[bi] = val
            bobs[bi] = min_req[mask, ri].mean() if mask.any() else 0.
        cdf = np.zeros_like(bins)                                         # _calcCDF :471-479
        cdf[0] = bins[0]
        for i in range(1, len(cdf)):
            cdf[i] = cdf[i - 1] + bins[i]
        cdf = np.zeros_like(bins) if cdf.max() == 0.0 else cdf / cdf.max()
        bins_l.append(bins); obs_l.append(bobs); cdf_l.append(cdf)
    bins_full = np.vstack(bins_l).T; cdf_full = np.vstack(cdf_l).T; obs_full = np.vstack(obs_l).T
    ddof = 1 if len(bins_full) > 1 else 0                                 # VectorResult :177-184 (len = #bins!)
    observ = np.zeros(bin_count)                                          # _setObservability :390-402
    for bi in range(bin_count):
        o = obs_full[bi, :]
        o = o[o < np.inf]
        if len(o):
            observ[bi] = o.max()
    # Moments :60-122
    vals = contribs[:, param_index, :]
    val = np.zeros(R); mu = np.zeros(R); var = np.zeros(R); skw = np.zeros(R); krt = np.zeros(R)
    for ri in range(R):
        valid = (vals[:, ri] > min(lower, upper)) * (vals[:, ri] < max(lower, upper))
        if not valid.any():
            continue
        rs, f = vals[valid, ri], fr[valid, ri]
        val[ri] = sum(f)
        mu[ri] = sum(rs * f)
        if 0 != sum(f):
            mu[ri] /= sum(f)
        var[ri] = sum((rs - mu[ri])**2 * f) / sum(f)
        sg = np.sqrt(abs(var[ri]))
        if (sum(f) * sg) == 0.0:
            continue
        skw[ri] = sum((rs - mu[ri])**3 * f) / (sum(f) * sg**3)
        krt[ri] = sum((rs - mu[ri])**4 * f) / (sum(f) * sg**4)
    md = 1 if R > 1 else 0
    moments = []
    for a in (val, mu, var, skw, krt):
        moments += [a.mean(), a.std(ddof=md)]
    return dict(edges=edges, bins_full=bins_full, bins_mean=bins_full.mean(axis=1),
                bins_std=bins_full.std(axis=1, ddof=ddof), cdf_mean=cdf_full.mean(axis=1),
                cdf_std=cdf_full.std(axis=1, ddof=ddof), observability=observ,
                moments=np.array(moments))


# ----------------------------------------------------------------------------- input preparation
def prepare_uncertainty(intensity, sigma_raw, fu_min):
    """DataObj._prepareUncertainty (dataobj/dataobj.py:204-227): uncertainties are raised to at least
    fu_min * intensity; without an uncertainty column that floor is used; non-finite entries become inf."""
    intensity = np.asarray(intensity, dtype=float)
    floor = fu_min * intensity
    if sigma_raw is None:
        upd = floor.copy()
    else:
        upd = np.maximum(np.asarray(sigma_raw, dtype=float), floor)
    upd[True ^ np.isfinite(upd)] = np.inf
    return upd


def rebin_edges(x, n_bin):
    """Bin edges of DataObj._reBin (dataobj/dataobj.py:312-316): log-spaced, last point included."""
    x = np.asarray(x, dtype=float)
    return np.logspace(np.log10(x.min()), np.log10(x.max() + np.diff(x)[-1] / 100.), n_bin + 1)


def rebin(x, f, fu, n_bin):
    """DataObj._reBin (dataobj/dataobj.py:288-345) on the sanitized vectors: per bin the mean of x and f,
    the larger of the standard error of the mean and the propagated uncertainty; empty bins dropped.
    Returns (x_binned, f_binned, fu_binned)."""
    x, f, fu = (np.asarray(a, dtype=float) for a in (x, f, fu))
    edges = rebin_edges(x, n_bin)
    xb = np.full(n_bin, np.nan); fb = np.full(n_bin, np.nan); ub = np.full(n_bin, np.nan)
    valid = np.zeros(n_bin, dtype=bool)
    for b in range(n_bin):
        m = (x >= edges[b]) & (x < edges[b + 1])
        cnt = int(m.sum())
        if cnt == 1:
            fb[b], ub[b], xb[b] = f[m][0], fu[m][0], x[m][0]
            valid[b] = True
        elif cnt > 1:
            fb[b], xb[b] = f[m].mean(), x[m].mean()
            valid[b] = True
            ub[b] = max(f[m].std(ddof=1) / np.sqrt(1. * cnt), np.sqrt((fu[m] ** 2).sum() / cnt))
    keep = (True ^ np.isnan(fb)) & valid
    return xb[keep], fb[keep], ub[keep]

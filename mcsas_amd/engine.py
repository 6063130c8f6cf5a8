"""Array-level front of the HIP library: builds `mcsas_problem` blocks and runs them.

This is the thin layer the class-level mirror (`mcsas_amd.mcsas.McSAS`, `mcsas_amd.scatteringmodels`)
and the reference-side binding shown in INTEGRATION.md both sit on.  Everything numerical happens
in libmcsas_hip.so; nothing here evaluates a form factor or a chi-squared on the CPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import MAX_ACTIVE, MAX_DEVICES, MAX_PARAMS, Problem, Result, as_dp, check, f64

MODEL_SPHERE, MODEL_CYL_ISO, MODEL_ELL_CS, MODEL_KHOLODENKO = 0, 1, 2, 3
MODEL_ELL_ISO, MODEL_SPH_CS, MODEL_GAUSS_CHAIN, MODEL_LMA_SPHERE = 4, 5, 6, 7
MODEL_PLUGIN0 = 64      # first id of a run-time model plug-in (include/mcsas_hip.h: MCSAS_MODEL_PLUGIN0)
MODEL_HOST = -1         # a model that exists only as host code (Python formfactor / volume): rows through analyse_host_rows


class PluginCompileError(ValueError):
    """A model plug-in's HIP source does not compile; .log holds the compiler output."""
    def __init__(self, message, log):
        super().__init__(message + "\n" + log)
        self.log = log


def release_cached_memory(tuning=False):
    """Returns the device / pinned memory parked by destroyed plans to the driver (mcsas_hip_release_cached_memory)."""
    check(_lib.load(tuning).mcsas_hip_release_cached_memory(), _lib.load(tuning))


def compile_plugin(source: str, tuning=False) -> int:
    """Registers HIP source text for a model outside the built-in ones (mcsas_hip_plugin_compile, include/mcsas_hip.h: the
    four functions mcsas_plugin_formfactor / _volume / _absvolume / _surface) and returns its model id.  Needs no GPU; the
    same text gives the same id.  This is what stands in for the reference's model discovery (utils/findmodels.py:120-186)."""
    lib = _lib.load(tuning)
    mid = C.c_int32(-1)
    rc = lib.mcsas_hip_plugin_compile(source.encode("utf-8"), C.byref(mid))
    if rc != 0:
        raise PluginCompileError(lib.mcsas_hip_last_error().decode("utf-8", "replace"),
                                 lib.mcsas_hip_plugin_log().decode("utf-8", "replace"))
    return int(mid.value)
EXEC_AUTO, EXEC_WAVE, EXEC_WORKGROUP, EXEC_PIPELINE = 0, 1, 2, 3
GEN_UNIFORM, GEN_EXP1, GEN_EXP2, GEN_EXP3 = 0, 1, 2, 3
INT64_MAX = (1 << 63) - 1


@dataclass
class ModelSetup:
    """A configured scattering model, flattened: full parameter vector (order of the reference's
    `parameters` tuple) plus, per active parameter, generator range/kind and clip range."""
    model_id: int
    params: np.ndarray                 # all parameter values (SI)
    active_index: tuple                # ascending indices into params
    gen_lo: np.ndarray                 # activeRange ∩ valueRange
    gen_hi: np.ndarray
    gen_kind: tuple
    clip_lo: np.ndarray                # valueRange
    clip_hi: np.ndarray
    start_value: np.ndarray = None     # startFromMinimum fill (mcsas.py:310-315)

    @property
    def n_active(self):
        return len(self.active_index)


@dataclass
class Settings:
    """The algorithm parameters of mcsas/mcsasparameters.json that reach the hot path."""
    n_contrib: int = 300
    n_reps: int = 10
    max_iter: int = 100000
    comp_exp: float = 0.6666666
    conv_crit: float = 1.0
    find_background: bool = True
    positive_background: bool = False
    start_from_minimum: bool = False
    max_retries: int = 5
    show_incomplete: bool = False
    # execution knobs (no reference counterpart)
    seed: int = 0
    rep_offset: int = 0
    device: int = -1                   # HIP device ordinal (-1: current)
    devices: tuple = ()                # several GPUs: analyse() runs contiguous blocks of repetitions on these at once
    waves_per_chain: int = 0
    cache_intensities: int = -1
    exec_mode: int = 0                 # MCSAS_EXEC_*: 0 auto, 1 wave, 2 workgroup, 3 pipeline
    debug_flags: int = 0               # tuning / ablation word: non-zero runs on the measurement build (libmcsas_hip_tuning.so)


def _fill(arr, values, n):
    for i in range(n):
        arr[i] = values[i]


class HipProblem:
    """Owns the numpy buffers a `mcsas_problem` points at (ctypes does not keep them alive)."""

    def __init__(self, model: ModelSetup, q, intensity, sigma, st: Settings, replay=None, stop=None, smear=None):
        if model.n_active < 0 or model.n_active > MAX_ACTIVE:
            raise ValueError("0..%d active parameters supported, got %d" % (MAX_ACTIVE, model.n_active))
        self.q, self.I, self.sigma = f64(q).ravel(), f64(intensity).ravel(), f64(sigma).ravel()
        if not (len(self.q) == len(self.I) == len(self.sigma)):
            raise ValueError("q, intensity and sigma must have the same length")
        self.model, self.st = model, st
        p = Problem()
        p.struct_size = C.sizeof(Problem)
        p.model_id = model.model_id
        p.nq = len(self.q)
        p.q, p.intensity, p.sigma = as_dp(self.q), as_dp(self.I), as_dp(self.sigma)
        _fill(p.params, list(model.params) + [0.0] * (MAX_PARAMS - len(model.params)), MAX_PARAMS)
        P = model.n_active
        p.n_active = P
        _fill(p.active_index, model.active_index, P)
        _fill(p.gen_lo, model.gen_lo, P); _fill(p.gen_hi, model.gen_hi, P)
        _fill(p.gen_kind, model.gen_kind, P)
        _fill(p.clip_lo, model.clip_lo, P); _fill(p.clip_hi, model.clip_hi, P)
        sv = model.start_value if model.start_value is not None else np.zeros(P)
        _fill(p.start_value, sv, P)
        p.n_contrib, p.n_reps = int(st.n_contrib), int(st.n_reps)
        p.max_iter = int(min(float(st.max_iter), float(INT64_MAX // 2)))
        p.comp_exp, p.conv_crit = float(st.comp_exp), float(st.conv_crit)
        p.max_retries = int(st.max_retries)
        p.find_background = int(bool(st.find_background))
        p.positive_background = int(bool(st.positive_background))
        p.start_from_minimum = int(bool(st.start_from_minimum))
        p.seed = int(st.seed) & 0xFFFFFFFFFFFFFFFF
        p.rep_offset = int(st.rep_offset)
        p.reserved0 = int(st.debug_flags)
        self.replay = None
        if replay is not None:
            self.replay = f64(replay).reshape(p.n_reps, -1)
            p.replay_stream = as_dp(self.replay)
            p.replay_len = self.replay.shape[1]
        self.stop = stop                      # ctypes c_int32 the caller may set to 1
        if stop is not None:
            p.stop = C.pointer(stop)
        p.device = int(st.device)
        devs = tuple(int(d) for d in (st.devices or ()))
        if len(devs) > MAX_DEVICES:
            raise ValueError("at most %d devices, got %d" % (MAX_DEVICES, len(devs)))
        p.n_devices = len(devs)
        _fill(p.devices, devs, len(devs))
        p.waves_per_chain = int(st.waves_per_chain)
        p.cache_intensities = int(st.cache_intensities)
        p.exec_mode = int(st.exec_mode)
        # beam-profile smearing: `smear` carries what SASConfig.prepareSmearing produced (locs[Q][K],
        # q_offset[K], weights[K]; dataobj.Smearing or anything with those attributes), or None
        self.smear = None
        if smear is not None and getattr(smear, "q_offset", None) is not None:
            locs = f64(smear.locs)
            if locs.ndim != 2 or locs.shape[0] != p.nq:
                raise ValueError("smear.locs must be (nq, K), got %r" % (locs.shape,))
            self.smear = (locs, f64(smear.q_offset).ravel(), f64(smear.weights).ravel())
            if not (locs.shape[1] == len(self.smear[1]) == len(self.smear[2])):
                raise ValueError("smear: locs, q_offset and weights disagree on K")
            p.smear_nk = locs.shape[1]
            p.smear_locs, p.smear_q_offset, p.smear_weights = (as_dp(a) for a in self.smear)
        self.c = p


class ChainResults:
    """numpy-backed `mcsas_result`."""

    def __init__(self, n_contrib, n_active, n_reps, nq):
        self.contribs = np.zeros((n_contrib, n_active, n_reps))
        self.fit = np.zeros((nq, n_reps))
        self.chisq = np.zeros(n_reps); self.scaling = np.zeros(n_reps); self.background = np.zeros(n_reps)
        self.num_iter = np.zeros(n_reps, dtype=np.int64); self.num_moves = np.zeros(n_reps, dtype=np.int64)
        self.attempts = np.zeros(n_reps, dtype=np.int32); self.converged = np.zeros(n_reps, dtype=np.int32)
        self.seconds = np.zeros(n_reps); self.draws = np.zeros(n_reps, dtype=np.int64)
        r = Result()
        r.struct_size = C.sizeof(Result)
        r.contribs, r.fit = as_dp(self.contribs), as_dp(self.fit)
        r.chisq, r.scaling, r.background = as_dp(self.chisq), as_dp(self.scaling), as_dp(self.background)
        r.num_iter = self.num_iter.ctypes.data_as(C.POINTER(C.c_int64))
        r.num_moves = self.num_moves.ctypes.data_as(C.POINTER(C.c_int64))
        r.attempts = self.attempts.ctypes.data_as(C.POINTER(C.c_int32))
        r.converged = self.converged.ctypes.data_as(C.POINTER(C.c_int32))
        r.seconds = as_dp(self.seconds)
        r.draws = self.draws.ctypes.data_as(C.POINTER(C.c_int64))
        self.c = r


def analyse(model: ModelSetup, q, intensity, sigma, st: Settings, replay=None, stop=None, smear=None) -> ChainResults:
    """All repetitions of McSAS.analyse (mcsas.py:214-262) in one kernel launch."""
    lib = _lib.load(tuning=bool(st.debug_flags))
    prob = HipProblem(model, q, intensity, sigma, st, replay, stop, smear)
    res = ChainResults(st.n_contrib, model.n_active, st.n_reps, len(prob.q))
    check(lib.mcsas_hip_analyse(C.byref(prob.c), C.byref(res.c)), lib)
    return res


def analyse_host_rows(model: ModelSetup, q, intensity, sigma, st: Settings, row_fn, replay=None, stop=None, window=64) -> ChainResults:
    """McSAS.analyse for a model that exists only as host code (mcsas_hip_analyse_host_rows, include/mcsas_hip.h): the library draws
    the proposals of the next `window` steps of every chain, `row_fn(pset[n][n_active]) -> rows[n][nq]` evaluates the caller's own
    calcIntensity for all of them (bases/model/scatteringmodel.py:90-99, sasmodel.py:46-79), the device does the rest of mcFit."""
    lib = _lib.load()
    prob = HipProblem(model, q, intensity, sigma, st, replay, stop, None)
    prob.c.model_id = MODEL_HOST
    nq, P = len(prob.q), model.n_active
    res = ChainResults(st.n_contrib, P, st.n_reps, nq)
    failure = []

    def cb(user, n, pset_p, rows_p):
        try:
            pset = np.ctypeslib.as_array(pset_p, shape=(n, P))
            out = np.ctypeslib.as_array(rows_p, shape=(n, nq))
            out[:, :] = np.asarray(row_fn(pset), dtype=np.float64).reshape(n, nq)
            return 0
        except BaseException as e:           # (an exception must not cross the C frame: hand it over and re-raise below)
            failure.append(e)
            return 1

    rc = lib.mcsas_hip_analyse_host_rows(C.byref(prob.c), _lib.RowsCallback(cb), None, int(window), C.byref(res.c))
    if failure:
        raise failure[0]
    check(rc, lib)
    return res


def analyse_many(problems, streams=2):
    """Independent analyses — a series of data sets (gui/calc.py:271-330 runs them one after the other) — side by side: every
    problem gets a plan of its own (creation is cheap: the library keeps destroyed plans' memory), the plans go round `streams`
    HIP streams, two per stream in flight, so the chip always has one analysis' scan-bound ticks beside another's producer-bound
    ones.  `problems`: sequence of (model: ModelSetup, q, intensity, sigma, settings) or of dicts with those keys
    plus optional replay / stop / smear.  Returns the ChainResults in order — each identical to analyse() of that problem."""
    problems = [p if isinstance(p, dict) else dict(model=p[0], q=p[1], intensity=p[2], sigma=p[3], st=p[4]) for p in problems]
    if not problems:
        return []
    lib = _lib.load(tuning=bool(problems[0]["st"].debug_flags))
    per_dev = {}                                              # device ordinal -> the stream handles created on THAT device

    def stream_for(dev, k):
        hs = per_dev.setdefault(dev, [])
        if len(hs) < max(1, streams):
            h = C.c_void_p()
            check(lib.mcsas_hip_stream_create(C.c_int32(dev), C.byref(h)), lib)
            hs.append(h)
        return hs[k % len(hs)]

    def direct(pr):
        return analyse(pr["model"], pr["q"], pr["intensity"], pr["sigma"], pr["st"], pr.get("replay"), pr.get("stop"), pr.get("smear"))

    out, pending, launched = [None] * len(problems), [], {}
    try:
        for i, pr in enumerate(problems):
            st = pr["st"]
            # what a plan cannot do goes through mcsas_hip_analyse at its place in the sequence: a model with no active parameter
            # (one contribution, nothing to fit: mcsas.py:198-199) and a device list (the library shards the repetitions itself)
            if pr["model"].n_active == 0 or st.devices:
                out[i] = direct(pr)
                continue
            cap = 2 * max(1, streams)
            while len(pending) >= cap:
                j, pl = pending.pop(0)
                out[j] = pl.fetch(); pl.close()
            pl = Plan(pr["model"], pr["q"], pr["intensity"], pr["sigma"], st, pr.get("replay"), pr.get("stop"), pr.get("smear"))
            k = launched.get(st.device, 0); launched[st.device] = k + 1
            pl.launch(stream=stream_for(st.device, k).value)
            pending.append((i, pl))
        while pending:
            j, pl = pending.pop(0)
            out[j] = pl.fetch(); pl.close()
    finally:
        for _, pl in pending:
            pl.close()
        for hs in per_dev.values():
            for h in hs:
                lib.mcsas_hip_stream_destroy(h)
    return out


class Plan:
    """Resident plan: data and workspaces stay in HBM; launch/fetch can be repeated (bench.py)."""

    def __init__(self, model: ModelSetup, q, intensity, sigma, st: Settings, replay=None, stop=None, smear=None):
        self.lib = _lib.load(tuning=bool(st.debug_flags))
        self.prob = HipProblem(model, q, intensity, sigma, st, replay, stop, smear)
        self.h = C.c_void_p()
        check(self.lib.mcsas_hip_plan_create(C.byref(self.prob.c), C.byref(self.h)), self.lib)

    def launch(self, stream=None, slot=0):
        """Enqueues one analysis.  `slot`: which of the plan's result sets (mcsas_hip.h: MCSAS_PLAN_SLOTS) it writes — launch
        into slot 1 while slot 0 is fetched and the device never waits for the host (same stream: the slots share the workspaces)."""
        check(self.lib.mcsas_hip_plan_launch_slot(self.h, C.c_void_p(stream or 0), C.c_int32(slot)), self.lib)

    def fetch(self, want_arrays=True, slot=0):
        st = self.prob.st
        res = ChainResults(st.n_contrib, self.prob.model.n_active, st.n_reps, len(self.prob.q)) if want_arrays else None
        check(self.lib.mcsas_hip_plan_fetch_slot(self.h, C.c_int32(slot), C.byref(res.c) if res is not None else None), self.lib)
        return res

    def reseed(self, seed, rep_offset=0):
        check(self.lib.mcsas_hip_plan_reseed(self.h, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_int32(rep_offset)), self.lib)

    @property
    def info(self):
        """dict: exec mode the library chose, waves per chain, q slots per lane, window, launches, cache."""
        v = (C.c_int32 * 8)()
        check(self.lib.mcsas_hip_plan_info(self.h, v), self.lib)
        names = {1: "wave", 2: "workgroup", 3: "pipeline"}
        return dict(exec_mode=names.get(v[0], str(v[0])), waves_per_chain=v[1], q_per_lane=v[2], window=v[3],
                    launches=v[4], cached_rows=bool(v[5]))

    @property
    def last_ms(self):
        v = C.c_double()
        check(self.lib.mcsas_hip_plan_last_ms(self.h, C.byref(v)), self.lib)
        return v.value

    @property
    def total_steps(self):
        v = C.c_int64()
        check(self.lib.mcsas_hip_plan_total_steps(self.h, C.byref(v)), self.lib)
        return v.value

    def close(self):
        if self.h:
            self.lib.mcsas_hip_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def model_calc(model: ModelSetup, q, pset, comp_exp, want_rows=False, device=-1, smear=None):
    """ScatteringModel.calc(data, pset, compensationExponent) (scatteringmodel.py:79-109) on the GPU.
    Returns cumInt, vset, wset, sset (and rows[n][Q] when asked)."""
    lib = _lib.load()
    q = f64(q).ravel()
    pset = f64(pset).reshape(-1, model.n_active)
    st = Settings(n_contrib=1, n_reps=1, comp_exp=comp_exp, device=device)
    prob = HipProblem(model, q, np.ones_like(q), np.ones_like(q), st, smear=smear)
    n = len(pset)
    cum = np.zeros(len(q)); vset = np.zeros(n); wset = np.zeros(n); sset = np.zeros(n)
    rows = np.zeros((n, len(q))) if want_rows else None
    check(lib.mcsas_hip_model_calc(C.byref(prob.c), as_dp(pset), n, as_dp(cum), as_dp(vset), as_dp(wset),
                                   as_dp(sset), as_dp(rows) if rows is not None else None))
    return (cum, vset, wset, sset, rows) if want_rows else (cum, vset, wset, sset)


def bgfit(intensity, sigma, model_int, find_background=True, positive_background=False, num_params=1, device=-1):
    """BackgroundScalingFit.calc (backgroundscalingfit.py:112-139): returns (sc[2], conval, aGoFs)."""
    lib = _lib.load()
    I, s, c = f64(intensity).ravel(), f64(sigma).ravel(), f64(model_int).ravel()
    out = np.zeros(4)
    check(lib.mcsas_hip_bgfit(len(I), as_dp(I), as_dp(s), as_dp(c), int(find_background), int(positive_background),
                              int(num_params), int(device), as_dp(out)))
    return out[:2].copy(), float(out[2]), float(out[3])


def observability(model: ModelSetup, q, sigma, contribs, scaling, vol_frac, comp_exp, device=-1, smear=None):
    """Per-contribution minimum visible volume fraction (mcsas.py:575-590) for all reps."""
    lib = _lib.load()
    contribs = f64(contribs)
    N, P, R = contribs.shape
    q = f64(q).ravel(); sigma = f64(sigma).ravel()
    st = Settings(n_contrib=N, n_reps=R, comp_exp=comp_exp, device=device)
    prob = HipProblem(model, q, np.ones_like(q), sigma, st, smear=smear)
    scaling = f64(scaling).ravel(); vol_frac = f64(vol_frac).reshape(N, R)
    out = np.zeros((N, R))
    check(lib.mcsas_hip_observability(C.byref(prob.c), as_dp(contribs), as_dp(scaling), as_dp(vol_frac), as_dp(out)))
    return out


def histogram_prep(model: ModelSetup, q, intensity, sigma, contribs, comp_exp, find_background=True,
                   positive_background=False, device=-1, smear=None):
    """First half of McSAS.histogram() (mcsas.py:549-594) for all repetitions in one call: returns
    scaling[2][R] (scale, background per rep), vset, wset, sset and the visibility limits, each [N][R]."""
    lib = _lib.load()
    contribs = f64(contribs)
    N, P, R = contribs.shape
    st = Settings(n_contrib=N, n_reps=R, comp_exp=comp_exp, device=device, find_background=find_background,
                  positive_background=positive_background)
    prob = HipProblem(model, q, intensity, sigma, st, smear=smear)
    sc = np.zeros((2, R)); v = np.zeros((N, R)); w = np.zeros((N, R)); s = np.zeros((N, R)); mv = np.zeros((N, R))
    check(lib.mcsas_hip_histogram_prep(C.byref(prob.c), as_dp(contribs), as_dp(sc), as_dp(v), as_dp(w), as_dp(s), as_dp(mv)))
    return sc, v, w, s, mv


HISTOGRAM_MAX_CONTRIBS = 4096     # mcsas_hip_histogram stages a repetition's contributions in LDS
YWEIGHT_INDEX = {"vol": 0, "num": 1, "int": 2, "surf": 3}


def histogram_device(model: ModelSetup, q, intensity, sigma, contribs, comp_exp, specs, find_background=True,
                     positive_background=False, device=-1, smear=None):
    """McSAS.histogram() (mcsas.py:445-615) for all repetitions on the device in one call (mcsas_hip_histogram).
    `specs`: per histogram a dict(param_index, yweight, edges (n_bin + 1 lower edges), lower, upper).
    Returns (scaling[2][R], fractions dict like mcsas.histogram's {'vol': (vf, mv), ...}, per histogram a dict with
    bins[n_bin][R], obs[n_bin][R], cdf[n_bin][R], moments[5][R])."""
    lib = _lib.load()
    contribs = f64(contribs)
    N, P, R = contribs.shape
    st = Settings(n_contrib=N, n_reps=R, comp_exp=comp_exp, device=device, find_background=find_background,
                  positive_background=positive_background)
    prob = HipProblem(model, q, intensity, sigma, st, smear=smear)
    arr = (_lib.HistogramSpec * max(len(specs), 1))()
    keep, n_out = [], 0
    for h, sp in enumerate(specs):
        e = f64(sp["edges"]).ravel()
        keep.append(e)
        arr[h].param_index = int(sp["param_index"]); arr[h].weighting = YWEIGHT_INDEX[sp["yweight"]]
        arr[h].n_bin = len(e) - 1; arr[h].lower = float(sp["lower"]); arr[h].upper = float(sp["upper"]); arr[h].edges = as_dp(e)
        n_out += 3 * (len(e) - 1) * R + 5 * R
    sc = np.zeros((2, R)); frac = np.zeros((8, N, R)); out = np.zeros(max(n_out, 1))
    check(lib.mcsas_hip_histogram(C.byref(prob.c), as_dp(contribs), len(specs), arr, as_dp(sc), as_dp(frac), as_dp(out)), lib)
    fractions = dict(vol=(frac[0], frac[4]), num=(frac[1], frac[5]), int=(frac[2], frac[6]), surf=(frac[3], frac[7]))
    hists, off = [], 0
    for sp, e in zip(specs, keep):
        nb = len(e) - 1
        blk = out[off:off + 3 * nb * R + 5 * R]
        hists.append(dict(bins=blk[:nb * R].reshape(nb, R), obs=blk[nb * R:2 * nb * R].reshape(nb, R),
                          cdf=blk[2 * nb * R:3 * nb * R].reshape(nb, R), moments=blk[3 * nb * R:].reshape(5, R)))
        off += 3 * nb * R + 5 * R
    return sc, fractions, hists


def prepare_uncertainty(intensity, sigma_raw, fu_min, device=-1):
    """DataObj._prepareUncertainty (dataobj/dataobj.py:204-227) on the GPU."""
    lib = _lib.load()
    I = f64(intensity).ravel()
    su = None if sigma_raw is None else f64(sigma_raw).ravel()
    out = np.zeros(len(I))
    check(lib.mcsas_hip_prepare_uncertainty(len(I), as_dp(I), as_dp(su) if su is not None else None, float(fu_min),
                                            int(device), as_dp(out)))
    return out


def rebin(x, f, fu, n_bin, device=-1):
    """DataObj._reBin (dataobj/dataobj.py:288-345) on the GPU: (x_binned, f_binned, fu_binned).  The
    log-spaced edges are the reference's numpy expression (:312-316), evaluated here on the host."""
    lib = _lib.load()
    x, f, fu = f64(x).ravel(), f64(f).ravel(), f64(fu).ravel()
    n_bin = int(n_bin)
    edges = np.logspace(np.log10(x.min()), np.log10(x.max() + np.diff(x)[-1] / 100.), n_bin + 1)
    xo, fo, uo = np.zeros(n_bin), np.zeros(n_bin), np.zeros(n_bin)
    k = C.c_int32(0)
    check(lib.mcsas_hip_rebin(len(x), as_dp(x), as_dp(f), as_dp(fu), n_bin, as_dp(edges), int(device),
                              as_dp(xo), as_dp(fo), as_dp(uo), C.byref(k)))
    return xo[:k.value].copy(), fo[:k.value].copy(), uo[:k.value].copy()


def device_count():
    return _lib.load().mcsas_hip_device_count()


def shard(n_reps, n_devices, index):
    """(first, count): the block of repetitions mcsas_hip_analyse gives device-list entry `index`."""
    first, count = C.c_int32(0), C.c_int32(0)
    check(_lib.load().mcsas_hip_shard(int(n_reps), int(n_devices), int(index), C.byref(first), C.byref(count)))
    return first.value, count.value

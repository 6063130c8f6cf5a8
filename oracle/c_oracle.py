"""ctypes front of oracle/c/libmcsas_oracle_c.so — TEST INFRASTRUCTURE ONLY (second CPU checker, compiled
CPU baseline for bench.py).  Never imported by the product (mcsas_amd/).  Models: Sphere, CylindersIsotropic,
EllipsoidalCoreShell (BASELINE configs 2-4)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "c", "libmcsas_oracle_c.so")
_dp, _i64p, _i32p = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
MAX_ACTIVE, MAX_PARAMS = 4, 8
MODELS = (0, 1, 2)                                         # oracle.mcsas_oracle SPHERE, CYL_ISO, ELL_CS


class _Problem(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("nq", C.c_int32), ("n_contrib", C.c_int32), ("n_reps", C.c_int32), ("max_retries", C.c_int32),
        ("n_active", C.c_int32),
        ("active", C.c_int32 * MAX_ACTIVE), ("gen", C.c_int32 * MAX_ACTIVE),
        ("q", _dp), ("intensity", _dp), ("sigma", _dp),
        ("gen_lo", C.c_double * MAX_ACTIVE), ("gen_hi", C.c_double * MAX_ACTIVE), ("start_value", C.c_double * MAX_ACTIVE),
        ("values", C.c_double * MAX_PARAMS), ("clip_lo", C.c_double * MAX_PARAMS), ("clip_hi", C.c_double * MAX_PARAMS),
        ("comp_exp", C.c_double), ("conv_crit", C.c_double),
        ("max_iter", C.c_int64),
        ("find_bg", C.c_int32), ("pos_bg", C.c_int32), ("start_from_min", C.c_int32), ("rep_offset", C.c_int32),
        ("seed", C.c_uint64),
        ("replay", _dp), ("replay_len", C.c_int64),
        ("contribs", _dp), ("fit", _dp), ("chisq", _dp), ("scaling", _dp), ("background", _dp),
        ("num_iter", _i64p), ("num_moves", _i64p), ("draws", _i64p), ("total_steps", _i64p),
        ("attempts", _i32p), ("converged", _i32p), ("overflow", _i32p),
        ("accepted", _i32p), ("accepted_cap", C.c_int64),
    ]


def build():
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")], stdout=subprocess.DEVNULL)


def load():
    src = os.path.join(_HERE, "c", "mcsas_oracle.c")
    if not os.path.exists(LIB) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB)):
        build()
    lib = C.CDLL(LIB)
    lib.mcsas_c_analyse.argtypes = [C.POINTER(_Problem), C.c_int]
    lib.mcsas_c_model_calc.argtypes = [C.POINTER(_Problem), _dp, C.c_int, _dp]
    lib.mcsas_c_philox_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    lib.mcsas_c_philox_uniform.restype = C.c_double
    lib.mcsas_c_j1.argtypes = [C.c_double]
    lib.mcsas_c_j1.restype = C.c_double
    return lib


class Result(object):
    pass


def _f(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _fill_model(p, spec, comp_exp, start_value=None):
    """`spec`: an oracle.mcsas_oracle.ModelSpec (model id, active columns, generator ranges and kinds, full parameter vector)."""
    from oracle import mcsas_oracle as O
    mid = int(spec.model_id)
    if mid not in MODELS:
        raise ValueError("the C oracle restates Sphere, CylindersIsotropic and EllipsoidalCoreShell only (model id %d)" % mid)
    P = spec.n_active
    p.model, p.n_active = mid, P
    for c in range(P):
        p.active[c], p.gen[c] = int(spec.active[c]), int(spec.gen[c])
        p.gen_lo[c], p.gen_hi[c] = float(spec.lo[c]), float(spec.hi[c])
        p.start_value[c] = float(start_value[c]) if start_value is not None else 0.0
    vr = O.PARAM_VALUE_RANGE[mid]
    for i, v in enumerate(spec.values):
        p.values[i], p.clip_lo[i], p.clip_hi[i] = float(v), float(vr[i][0]), float(vr[i][1])
    p.comp_exp = float(comp_exp)


def analyse(spec, q, I, sigma, n_contrib, n_reps, max_iter, conv_crit, comp_exp=0.6666666,
            find_bg=True, pos_bg=False, start_from_min=False, start_value=None, max_retries=0, seed=0,
            rep_offset=0, replay=None, threads=1, want_accepted=0):
    """McSAS.analyse (mcsas.py:191-285) for the model `spec` describes, repetitions spread over `threads` host threads;
    random numbers from the Philox stream (seed, rep_offset + repetition) or from `replay` [n_reps][draws]."""
    lib = load()
    q, I, sigma = _f(q), _f(I), _f(sigma)
    R, N, Q, P = int(n_reps), int(n_contrib), len(q), spec.n_active
    r = Result()
    r.contribs = np.zeros((N, P, R)); r.fit = np.zeros((Q, R))
    r.chisq = np.zeros(R); r.scaling = np.zeros(R); r.background = np.zeros(R)
    r.num_iter = np.zeros(R, np.int64); r.num_moves = np.zeros(R, np.int64); r.draws = np.zeros(R, np.int64)
    r.total_steps = np.zeros(R, np.int64)
    r.attempts = np.zeros(R, np.int32); r.converged = np.zeros(R, np.int32); r.overflow = np.zeros(R, np.int32)
    r.accepted = np.full((R, max(1, int(want_accepted))), -1, np.int32)
    p = _Problem()
    _fill_model(p, spec, comp_exp, start_value)
    p.nq, p.n_contrib, p.n_reps, p.max_retries = Q, N, R, int(max_retries)
    p.q, p.intensity, p.sigma = (a.ctypes.data_as(_dp) for a in (q, I, sigma))
    p.conv_crit = float(conv_crit)
    p.max_iter = int(max_iter)
    p.find_bg, p.pos_bg, p.start_from_min, p.rep_offset = int(find_bg), int(pos_bg), int(start_from_min), int(rep_offset)
    p.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    keep = None
    if replay is not None:
        keep = _f(replay).reshape(R, -1)
        p.replay, p.replay_len = keep.ctypes.data_as(_dp), keep.shape[1]
    p.contribs, p.fit = r.contribs.ctypes.data_as(_dp), r.fit.ctypes.data_as(_dp)
    p.chisq, p.scaling, p.background = (a.ctypes.data_as(_dp) for a in (r.chisq, r.scaling, r.background))
    p.num_iter, p.num_moves, p.draws, p.total_steps = (a.ctypes.data_as(_i64p) for a in (r.num_iter, r.num_moves, r.draws, r.total_steps))
    p.attempts, p.converged, p.overflow = (a.ctypes.data_as(_i32p) for a in (r.attempts, r.converged, r.overflow))
    if want_accepted:
        p.accepted, p.accepted_cap = r.accepted.ctypes.data_as(_i32p), int(want_accepted)
    rc = lib.mcsas_c_analyse(C.byref(p), int(threads))
    if rc:
        raise RuntimeError("mcsas_c_analyse failed: %d" % rc)
    return r


def model_calc(spec, q, pset, comp_exp=0.6666666):
    """ScatteringModel.calc (scatteringmodel.py:79-105): the summed intensity of the parameter sets `pset` [n][n_active]."""
    lib = load()
    q = _f(q)
    pset = _f(pset).reshape(-1, spec.n_active)
    p = _Problem()
    _fill_model(p, spec, comp_exp)
    p.nq = len(q)
    p.q = q.ctypes.data_as(_dp)
    cum = np.zeros(len(q))
    lib.mcsas_c_model_calc(C.byref(p), pset.ctypes.data_as(_dp), len(pset), cum.ctypes.data_as(_dp))
    return cum


def j1(x):
    lib = load()
    return np.array([lib.mcsas_c_j1(float(v)) for v in np.atleast_1d(x)])


def analyse_sphere(q, I, sigma, lo, hi, n_contrib, n_reps, max_iter, conv_crit, comp_exp=0.6666666,
                   find_bg=True, pos_bg=False, start_from_min=False, start_value=0.0, max_retries=0, seed=0,
                   rep_offset=0, replay=None, clip=(0.0, np.inf), threads=1, want_accepted=0):
    """McSAS.analyse for the Sphere model on `threads` host threads; arguments as in oracle.mcsas_oracle."""
    from oracle import mcsas_oracle as O
    spec = O.ModelSpec.make("sphere", ["radius"], [lo], [hi])
    spec.lo[0], spec.hi[0] = float(lo), float(hi)
    return analyse(spec, q, I, sigma, n_contrib, n_reps, max_iter, conv_crit, comp_exp=comp_exp, find_bg=find_bg, pos_bg=pos_bg,
                   start_from_min=start_from_min, start_value=[start_value], max_retries=max_retries, seed=seed, rep_offset=rep_offset,
                   replay=replay, threads=threads, want_accepted=want_accepted)

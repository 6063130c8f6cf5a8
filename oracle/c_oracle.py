"""ctypes front of oracle/c/libmcsas_oracle_c.so — TEST INFRASTRUCTURE ONLY (second CPU checker, compiled
CPU baseline for bench.py).  Never imported by the product (mcsas_amd/).  Sphere model only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "c", "libmcsas_oracle_c.so")
_dp, _i64p, _i32p = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)


class _Problem(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("n_contrib", C.c_int32), ("n_reps", C.c_int32), ("max_retries", C.c_int32),
        ("q", _dp), ("intensity", _dp), ("sigma", _dp),
        ("gen_lo", C.c_double), ("gen_hi", C.c_double), ("clip_lo", C.c_double), ("clip_hi", C.c_double),
        ("sld", C.c_double), ("comp_exp", C.c_double), ("conv_crit", C.c_double), ("start_value", C.c_double),
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
    if not os.path.exists(LIB):
        build()
    lib = C.CDLL(LIB)
    lib.mcsas_c_analyse.argtypes = [C.POINTER(_Problem), C.c_int]
    lib.mcsas_c_philox_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    lib.mcsas_c_philox_uniform.restype = C.c_double
    return lib


class Result(object):
    pass


def analyse_sphere(q, I, sigma, lo, hi, n_contrib, n_reps, max_iter, conv_crit, comp_exp=0.6666666,
                   find_bg=True, pos_bg=False, start_from_min=False, start_value=0.0, max_retries=0, seed=0,
                   rep_offset=0, replay=None, clip=(0.0, np.inf), threads=1, want_accepted=0):
    """McSAS.analyse for the Sphere model on `threads` host threads; arguments as in oracle.mcsas_oracle."""
    lib = load()
    f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    q, I, sigma = f(q), f(I), f(sigma)
    R, N, Q = int(n_reps), int(n_contrib), len(q)
    r = Result()
    r.contribs = np.zeros((N, 1, R)); r.fit = np.zeros((Q, R))
    r.chisq = np.zeros(R); r.scaling = np.zeros(R); r.background = np.zeros(R)
    r.num_iter = np.zeros(R, np.int64); r.num_moves = np.zeros(R, np.int64); r.draws = np.zeros(R, np.int64)
    r.total_steps = np.zeros(R, np.int64)
    r.attempts = np.zeros(R, np.int32); r.converged = np.zeros(R, np.int32); r.overflow = np.zeros(R, np.int32)
    r.accepted = np.full((R, max(1, int(want_accepted))), -1, np.int32)
    p = _Problem()
    p.nq, p.n_contrib, p.n_reps, p.max_retries = Q, N, R, int(max_retries)
    p.q, p.intensity, p.sigma = (a.ctypes.data_as(_dp) for a in (q, I, sigma))
    p.gen_lo, p.gen_hi, p.clip_lo, p.clip_hi = float(lo), float(hi), float(clip[0]), float(clip[1])
    p.comp_exp, p.conv_crit, p.start_value = float(comp_exp), float(conv_crit), float(start_value)
    p.max_iter = int(max_iter)
    p.find_bg, p.pos_bg, p.start_from_min, p.rep_offset = int(find_bg), int(pos_bg), int(start_from_min), int(rep_offset)
    p.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    keep = None
    if replay is not None:
        keep = f(replay).reshape(R, -1)
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

"""ctypes binding of libmcsas_hip.so (include/mcsas_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded this module raises
and every product entry point fails loudly.  Build it with `python __graft_entry__.py build` (or
`make -C mcsas_amd/csrc`); hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

MAX_ACTIVE = 4
MAX_PARAMS = 8
ABI_VERSION = 4
MAX_DEVICES = 16

# MCSAS_HIP_LIB selects another build of the SAME library (e.g. the -DMCSAS_STAMPS diagnostic build)
LIB_PATH = os.environ.get("MCSAS_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmcsas_hip.so")

# every symbol include/mcsas_hip.h declares (tests check that the library exports all of them)
SYMBOLS = (
    "mcsas_hip_analyse", "mcsas_hip_analyse_host_rows", "mcsas_hip_shard", "mcsas_hip_plan_create", "mcsas_hip_plan_launch", "mcsas_hip_plan_fetch",
    "mcsas_hip_plan_launch_slot", "mcsas_hip_plan_fetch_slot",
    "mcsas_hip_plan_last_ms", "mcsas_hip_plan_total_steps", "mcsas_hip_plan_reseed", "mcsas_hip_plan_info",
    "mcsas_hip_plan_destroy", "mcsas_hip_model_calc", "mcsas_hip_bgfit", "mcsas_hip_observability",
    "mcsas_hip_histogram_prep", "mcsas_hip_histogram", "mcsas_hip_prepare_uncertainty", "mcsas_hip_rebin",
    "mcsas_hip_plugin_compile", "mcsas_hip_plugin_log", "mcsas_hip_release_cached_memory", "mcsas_hip_stream_create", "mcsas_hip_stream_destroy",
    "mcsas_hip_device_count", "mcsas_hip_abi_version", "mcsas_hip_is_tuning_build", "mcsas_hip_last_error",
)

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)


# mcsas_rows_callback (include/mcsas_hip.h): int cb(void *user, int32_t n, const double *pset, double *rows)
RowsCallback = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, _dp, _dp)


class HistogramSpec(C.Structure):
    """mcsas_histogram_spec (include/mcsas_hip.h)."""
    _fields_ = [("param_index", C.c_int32), ("weighting", C.c_int32), ("n_bin", C.c_int32), ("reserved", C.c_int32),
                ("lower", C.c_double), ("upper", C.c_double), ("edges", _dp)]


class Problem(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("model_id", C.c_int32),
        ("nq", C.c_int32),
        ("q", _dp), ("intensity", _dp), ("sigma", _dp),
        ("params", C.c_double * MAX_PARAMS),
        ("n_active", C.c_int32),
        ("active_index", C.c_int32 * MAX_ACTIVE),
        ("gen_lo", C.c_double * MAX_ACTIVE), ("gen_hi", C.c_double * MAX_ACTIVE),
        ("gen_kind", C.c_int32 * MAX_ACTIVE),
        ("clip_lo", C.c_double * MAX_ACTIVE), ("clip_hi", C.c_double * MAX_ACTIVE),
        ("start_value", C.c_double * MAX_ACTIVE),
        ("n_contrib", C.c_int32), ("n_reps", C.c_int32),
        ("max_iter", C.c_int64),
        ("comp_exp", C.c_double), ("conv_crit", C.c_double),
        ("max_retries", C.c_int32), ("find_background", C.c_int32),
        ("positive_background", C.c_int32), ("start_from_minimum", C.c_int32),
        ("seed", C.c_uint64),
        ("rep_offset", C.c_int32), ("reserved0", C.c_int32),
        ("replay_stream", _dp), ("replay_len", C.c_int64),
        ("stop", _i32p),
        ("device", C.c_int32), ("waves_per_chain", C.c_int32),
        ("cache_intensities", C.c_int32), ("exec_mode", C.c_int32),
        ("smear_nk", C.c_int32), ("reserved1", C.c_int32),
        ("smear_locs", _dp), ("smear_q_offset", _dp), ("smear_weights", _dp),
        ("n_devices", C.c_int32), ("devices", C.c_int32 * MAX_DEVICES), ("reserved2", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("reserved", C.c_uint32),
        ("contribs", _dp), ("fit", _dp), ("chisq", _dp), ("scaling", _dp), ("background", _dp),
        ("num_iter", _i64p), ("num_moves", _i64p), ("attempts", _i32p), ("converged", _i32p),
        ("seconds", _dp), ("draws", _i64p),
    ]


class McSASHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libmcsas_hip error %d: %s" % (code, message))
        self.code = code


_libs = {}
TUNING_LIB_PATH = os.environ.get("MCSAS_HIP_TUNING_LIB") or os.path.join(os.path.dirname(LIB_PATH), "libmcsas_hip_tuning.so")


def load(tuning=False):
    """Returns the loaded library; raises if it is not built (no fallback by design).  `tuning`: the measurement build
    (csrc/Makefile: -DMCSAS_TUNING) in which mcsas_problem.reserved0 selects variants of the pipeline mode — tools and the
    variant tests only; the release library refuses a non-zero reserved0."""
    key = bool(tuning)
    if key in _libs:
        return _libs[key]
    path = TUNING_LIB_PATH if tuning else LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            "mcsas_amd: %s not found. The HIP library is the product path and has no CPU fallback; "
            "build it with `python __graft_entry__.py build`." % path)
    lib = C.CDLL(path)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ImportError("%s lacks symbols: %s" % (os.path.basename(path), ", ".join(missing)))
    lib.mcsas_hip_abi_version.restype = C.c_int
    if lib.mcsas_hip_abi_version() != ABI_VERSION:
        raise ImportError("%s ABI %d, binding expects %d" % (os.path.basename(path), lib.mcsas_hip_abi_version(), ABI_VERSION))
    lib.mcsas_hip_is_tuning_build.restype = C.c_int
    if bool(lib.mcsas_hip_is_tuning_build()) != key and not os.environ.get("MCSAS_HIP_LIB"):
        raise ImportError("%s: %s build expected" % (os.path.basename(path), "tuning" if tuning else "release"))
    lib.mcsas_hip_last_error.restype = C.c_char_p
    lib.mcsas_hip_device_count.restype = C.c_int
    lib.mcsas_hip_analyse.argtypes = [C.POINTER(Problem), C.POINTER(Result)]
    lib.mcsas_hip_analyse_host_rows.argtypes = [C.POINTER(Problem), RowsCallback, C.c_void_p, C.c_int32, C.POINTER(Result)]
    lib.mcsas_hip_shard.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p]
    lib.mcsas_hip_plan_create.argtypes = [C.POINTER(Problem), C.POINTER(C.c_void_p)]
    lib.mcsas_hip_plan_launch.argtypes = [C.c_void_p, C.c_void_p]
    lib.mcsas_hip_plan_fetch.argtypes = [C.c_void_p, C.POINTER(Result)]
    lib.mcsas_hip_plan_launch_slot.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.mcsas_hip_plan_fetch_slot.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Result)]
    lib.mcsas_hip_plan_last_ms.argtypes = [C.c_void_p, _dp]
    lib.mcsas_hip_plan_total_steps.argtypes = [C.c_void_p, _i64p]
    lib.mcsas_hip_plan_reseed.argtypes = [C.c_void_p, C.c_uint64, C.c_int32]
    lib.mcsas_hip_plan_info.argtypes = [C.c_void_p, _i32p]
    lib.mcsas_hip_plan_destroy.argtypes = [C.c_void_p]
    lib.mcsas_hip_plan_destroy.restype = None
    lib.mcsas_hip_model_calc.argtypes = [C.POINTER(Problem), _dp, C.c_int32, _dp, _dp, _dp, _dp, _dp]
    lib.mcsas_hip_bgfit.argtypes = [C.c_int32, _dp, _dp, _dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp]
    lib.mcsas_hip_observability.argtypes = [C.POINTER(Problem), _dp, _dp, _dp, _dp]
    lib.mcsas_hip_histogram_prep.argtypes = [C.POINTER(Problem), _dp, _dp, _dp, _dp, _dp, _dp]
    lib.mcsas_hip_histogram.argtypes = [C.POINTER(Problem), _dp, C.c_int32, C.POINTER(HistogramSpec), _dp, _dp, _dp]
    lib.mcsas_hip_prepare_uncertainty.argtypes = [C.c_int32, _dp, _dp, C.c_double, C.c_int32, _dp]
    lib.mcsas_hip_rebin.argtypes = [C.c_int32, _dp, _dp, _dp, C.c_int32, _dp, C.c_int32, _dp, _dp, _dp, _i32p]
    lib.mcsas_hip_plugin_compile.argtypes = [C.c_char_p, _i32p]
    lib.mcsas_hip_stream_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    lib.mcsas_hip_stream_destroy.argtypes = [C.c_void_p]
    lib.mcsas_hip_stream_destroy.restype = None
    lib.mcsas_hip_plugin_log.restype = C.c_char_p
    _libs[key] = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        raise McSASHipError(rc, (lib or load()).mcsas_hip_last_error().decode("utf-8", "replace"))


def as_dp(a):
    return a.ctypes.data_as(_dp)


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))

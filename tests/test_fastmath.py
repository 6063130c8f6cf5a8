"""Accuracy of mcsas_amd/csrc/fastmath.h (the fp64 sincos / J1 / division the form-factor kernels use instead of
the device libm), measured on a HOST build of the same header: tests/native/fastmath_host.cpp compiled with g++
(-mfma so that fma() is the single-rounding instruction the GPU executes, -ffp-contract=off so that nothing else
is fused).  References: x87 extended precision (64-bit significand) for sin / cos / division / rsqrt, scipy's
Cephes J1 for the Bessel function.  The bounds asserted here are the ones fastmath.h states.

What does NOT carry over from the host: the hardware reciprocal / reciprocal-square-root seeds of div_fast and
rsqrt_fast (v_rcp_f64, v_rsq_f64) are replaced by float-precision stand-ins — both are refined by Newton steps
whose result does not depend on the seed's last bits; the device versions are exercised by the GPU parity tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "fastmath_host.cpp")
dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fastmath") / "libfastmath_host.so")
    cmd = ["g++", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", out, SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("host build of fastmath.h failed (no FMA on this CPU?): " + r.stderr[-300:])
    L = C.CDLL(out)
    if L.fm_long_double_digits() < 64:
        pytest.skip("long double is not extended precision on this host")
    return L


def ulp_of(ref):
    return np.spacing(np.abs(ref))


def sincos(lib, fn, x):
    s, c = np.empty_like(x), np.empty_like(x)
    getattr(lib, fn)(len(x), P(x), P(s), P(c))
    return s, c


def ref_sincos(lib, x):
    a, b, c, d = (np.empty_like(x) for _ in range(4))
    lib.fm_ref_sincos(len(x), P(x), P(a), P(b), P(c), P(d))
    return a, b, c, d


def point_sets():
    rs = np.random.RandomState(1)
    k = np.arange(1, 667000, dtype=float)                      # k pi/2 < 2^20
    near = k * (np.pi / 2)
    return {
        "uniform": rs.uniform(-2.0**20, 2.0**20, 1500000),
        "log": 10 ** rs.uniform(-8, np.log10(2.0**20 * 0.999999), 1500000),
        "small": rs.uniform(-4, 4, 500000),
        "tiny": np.concatenate([[0.0, 1e-300, -1e-300, 5e-324], 10 ** rs.uniform(-300, -8, 1000)]),
        # the doubles next to multiples of pi/2, where the reduced argument is smallest
        "near_kpi2": np.concatenate([near, np.nextafter(near, np.inf), np.nextafter(near, -np.inf)]),
        "handoff": np.nextafter(2.0**20, 0) - np.arange(0, 2000) * 2.0**-32,
    }


@pytest.mark.parametrize("name", list(point_sets()))
def test_sincos_fast_relative_error(lib, name):
    """sincos_fast, |x| < 2^20: three-term Cody-Waite + fdlibm kernels: <= 1.6 ulp everywhere (0.8 ulp of the
    kernels + the rounding of the reduced argument), including next to multiples of pi/2."""
    x = np.ascontiguousarray(point_sets()[name])
    sh, sl, ch, cl = ref_sincos(lib, x)
    s, c = sincos(lib, "fm_sincos_fast", x)
    es, ec = np.abs((s - sh) - sl), np.abs((c - ch) - cl)
    assert (es / ulp_of(sh)).max() <= 1.6 and (ec / ulp_of(ch)).max() <= 1.6
    assert es.max() <= 1.8e-16 and ec.max() <= 1.8e-16


@pytest.mark.parametrize("name", list(point_sets()))
def test_sincos_core_absolute_error(lib, name):
    """sincos_core (branch-free, two-term reduction, what the row kernels call when q r < 2^20): ABSOLUTE error
    <= 1.8e-16 everywhere and <= 1.6 ulp away from multiples of pi/2; next to them (|value| < 1e-9) the dropped
    third reduction term (< 2e-27) shows as a relative error, the absolute error there is <= 2e-26."""
    x = np.ascontiguousarray(point_sets()[name])
    sh, sl, ch, cl = ref_sincos(lib, x)
    s, c = sincos(lib, "fm_sincos_core", x)
    es, ec = np.abs((s - sh) - sl), np.abs((c - ch) - cl)
    assert es.max() <= 1.8e-16 and ec.max() <= 1.8e-16
    big_s, big_c = np.abs(sh) > 1e-9, np.abs(ch) > 1e-9
    assert (es[big_s] / ulp_of(sh[big_s])).max() <= 1.6 and (ec[big_c] / ulp_of(ch[big_c])).max() <= 1.6
    if (~big_s).any():
        assert es[~big_s].max() <= 2e-26
    if (~big_c).any():
        assert ec[~big_c].max() <= 2e-26


def test_sincos_large_arguments_take_libm(lib):
    """|x| >= 2^20 (and NaN / Inf) leave the fast path: libm on the host, ocml on the device."""
    x = np.array([2.0**20, -2.0**20, 3.0e9, 1e15, 1e300])
    s, c = sincos(lib, "fm_sincos_fast", x)
    np.testing.assert_allclose(s, np.sin(x), rtol=0, atol=2.3e-16)
    np.testing.assert_allclose(c, np.cos(x), rtol=0, atol=2.3e-16)
    s, c = sincos(lib, "fm_sincos_fast", np.array([np.nan, np.inf]))
    assert np.isnan(s).all() and np.isnan(c).all()


def test_j1_against_scipy(lib):
    """j1_fast / j1_core are the Cephes rationals behind scipy.special.j1 re-associated to one division per branch:
    5e-16 absolute against scipy over (0, 2^20), both branches and the x = 5 seam; j1_fast is odd."""
    from scipy.special import j1
    rs = np.random.RandomState(2)
    x = np.concatenate([rs.uniform(0, 5, 400000), rs.uniform(5, 60, 400000), 10 ** rs.uniform(-6, 6, 400000),
                        np.nextafter(5.0, 0) - np.arange(100) * 1e-15, 5.0 + np.arange(100) * 1e-15,
                        [3.8317059702075125, 7.015586669815619, 1e-12, 1048575.9]])
    x = np.ascontiguousarray(x[(x > 0) & (x < 2.0**20)])
    ref = j1(x)
    for fn in ("fm_j1_fast", "fm_j1_core"):
        y = np.empty_like(x)
        getattr(lib, fn)(len(x), P(x), P(y))
        assert np.abs(y - ref).max() <= 5e-16, fn
        rel = np.abs(y - ref)[np.abs(ref) > 1e-3] / np.abs(ref)[np.abs(ref) > 1e-3]
        assert rel.max() <= 3e-13, fn                       # relative: limited by the zeros of J1, as scipy's own
    xm = np.ascontiguousarray(-x[:1000])
    y = np.empty_like(xm)
    lib.fm_j1_fast(len(xm), P(xm), P(y))
    yp = np.empty_like(xm)
    xp = np.ascontiguousarray(x[:1000])
    lib.fm_j1_fast(len(xp), P(xp), P(yp))
    np.testing.assert_array_equal(y, -yp)


def test_div_and_rsqrt(lib):
    """div_fast: <= 1 ulp for normal-range operands; rsqrt_fast: <= 1.5 ulp.  (Operand range of the sweep: what
    the host stand-in for the hardware seed — a float — can hold; the kernels divide by x^3 and by polynomial
    denominators of x = q r in 1e-4 .. 1e6.)"""
    rs = np.random.RandomState(3)
    a = np.ascontiguousarray(rs.uniform(-1, 1, 1000000) * 10 ** rs.uniform(-100, 100, 1000000))
    b = np.ascontiguousarray(rs.uniform(0.5, 1, 1000000) * 10 ** rs.uniform(-30, 30, 1000000) * rs.choice([-1, 1], 1000000))
    y, hi, lo = np.empty_like(a), np.empty_like(a), np.empty_like(a)
    lib.fm_div_fast(len(a), P(a), P(b), P(y))
    lib.fm_ref_div(len(a), P(a), P(b), P(hi), P(lo))
    assert (np.abs((y - hi) - lo) / ulp_of(hi)).max() <= 1.0
    x = np.ascontiguousarray(10 ** rs.uniform(-30, 30, 1000000))
    lib.fm_rsqrt_fast(len(x), P(x), P(y))
    lib.fm_ref_rsqrt(len(x), P(x), P(hi), P(lo))
    assert (np.abs((y - hi) - lo) / ulp_of(hi)).max() <= 1.5


def test_expm1_neg_fast(lib):
    """expm1_neg_fast (x <= 0; the Kholodenko quadrature's one transcendental per point besides sincos): <= 2 ulp
    against x87 expm1l from -1e-300 down to -60, including the arguments next to the k ln2 / 2 seams and the tiny
    ones where exp(x) - 1 would cancel."""
    rs = np.random.RandomState(4)
    seams = (np.arange(1, 80) * 0.5 * np.log(2.0))
    x = -np.concatenate([rs.uniform(0, 2.5, 600000), 10 ** rs.uniform(-300, 0, 200000), rs.uniform(2, 60, 200000),
                         seams, np.nextafter(seams, 0), np.nextafter(seams, 100), [0.0, 1e-320, 745.0, 800.0]])
    x = np.ascontiguousarray(x)
    y, hi, lo = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    lib.fm_expm1_neg_fast(len(x), P(x), P(y))
    lib.fm_ref_expm1(len(x), P(x), P(hi), P(lo))
    err = np.abs((y - hi) - lo) / np.maximum(ulp_of(hi), 5e-324)
    assert err.max() <= 2.0, (err.max(), x[np.argmax(err)])

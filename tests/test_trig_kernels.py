"""CPU: the bounded-range sin / cos / tan and the atan2 / atan / asin of csrc/rtus_trig.h (what the forward trace's angle arithmetic runs on the
GPU) against 50-digit mpmath values.  The header is plain C++, so the SAME text is compiled here for the host
(g++ -mfma: fma() is the hardware instruction on both sides)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

mp = pytest.importorskip("mpmath")
SRC = os.path.join(ROOT, "tests", "native", "trig_host.cpp")
LIB = os.path.join(ROOT, "tests", "native", "libtrig_host.so")


@pytest.fixture(scope="module")
def lib():
    hdr = os.path.join(ROOT, "ray-tracing-ultrasound_amd", "csrc", "rtus_trig.h")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", LIB], check=True)
    L = C.CDLL(LIB)
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    for f in (L.t_sin, L.t_cos, L.t_tan, L.t_asin, L.t_atan):
        f.argtypes = [dp, C.c_int, dp]
    L.t_atan2.argtypes = [dp, dp, C.c_int, dp]
    L.t_err_asin.argtypes = [dp, C.c_int]
    L.t_err_asin.restype = C.c_double
    L.t_err_atan2.argtypes = [dp, dp, C.c_int]
    L.t_err_atan2.restype = C.c_double
    return L


def _call(f, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    o = np.empty_like(x)
    f(x, x.size, o)
    return o


def _ulp_err(got, x, fn):
    mp.mp.dps = 50
    worst = 0.0
    for g, v in zip(got, x):
        ref = fn(mp.mpf(float(v)))
        rf = float(ref)
        ulp = np.spacing(abs(rf)) if rf != 0 else 5e-324
        worst = max(worst, float(abs(mp.mpf(float(g)) - ref) / mp.mpf(float(ulp))))
    return worst


def test_sin_cos_tan_within_two_ulp_on_the_trace_range(lib):
    rng = np.random.default_rng(0)
    # every angle of the trace is a short sum of atan2 / atan / asin results and +-pi/2: |x| < 8
    x = np.concatenate([rng.uniform(-8.0, 8.0, 3000), rng.uniform(-0.9, 0.9, 1000), rng.normal(0, 1e-3, 200),
                        np.arange(-5, 6) * (np.pi / 2) + rng.normal(0, 1e-9, 11), [0.0, -0.0, 1e-300, np.pi / 4, -np.pi / 4]])
    assert _ulp_err(_call(lib.t_sin, x), x, mp.sin) < 1.0
    assert _ulp_err(_call(lib.t_cos, x), x, mp.cos) < 1.0
    assert _ulp_err(_call(lib.t_tan, x), x, mp.tan) < 2.0


def test_exactly_vertical_angles_and_specials(lib):
    """tan(fl(pi/2)) etc.: the reference's exactly vertical rays are such angles (main_rt.py:348, 375, 401); the
    reduction keeps 118 bits of pi/2, so they come out as the libraries give them."""
    x = np.array([np.pi / 2, -np.pi / 2, 3 * np.pi / 2, np.pi, -np.pi, 2 * np.pi])
    got = _call(lib.t_tan, x)
    assert np.array_equal(got, np.tan(x)) or np.max(np.abs(got - np.tan(x)) / np.abs(np.tan(x))) < 4e-16
    assert got[0] == 1.633123935319537e16
    s = _call(lib.t_sin, x)
    assert np.max(np.abs(s - np.sin(x))) < 1e-31 + 2.3e-16 * np.max(np.abs(np.sin(x)))
    for f in (lib.t_sin, lib.t_cos, lib.t_tan):
        o = _call(f, np.array([np.nan, np.inf, -np.inf]))
        assert np.isnan(o).all()
    assert _call(lib.t_sin, np.array([-0.0, 0.0])).tolist() == [0.0, 0.0]      # (the sign of a zero is not kept: nothing in the trace reads it)


def _atan2(lib, y, x):
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    o = np.empty_like(y)
    lib.t_atan2(y, x, y.size, o)
    return o


def test_inverse_functions_against_mpmath(lib):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-1, 1, 2000), rng.uniform(-0.5, 0.5, 500), 1 - 10 ** rng.uniform(-16, -0.3, 500),
                        [0.5, -0.5, 0.4999999999999999, 0.975, 1e-9, 1e-300]])
    assert _ulp_err(_call(lib.t_asin, x), x, mp.asin) < 1.0
    x = np.concatenate([rng.uniform(-3, 3, 2000), 10 ** rng.uniform(-12, 12, 500), [0.41421356237309503, 0.4142135623730951, 1.0, 2.414213562373095]])
    assert _ulp_err(_call(lib.t_atan, x), x, mp.atan) < 1.7
    mp.mp.dps = 50
    y, xx = rng.normal(size=3000), rng.normal(size=3000)
    got = _atan2(lib, y, xx)
    worst = max(float(abs(mp.mpf(float(g)) - mp.atan2(mp.mpf(float(a)), mp.mpf(float(b)))) / mp.mpf(float(np.spacing(abs(g)))))
                for g, a, b in zip(got, y, xx))
    assert worst < 1.7


def test_inverse_functions_bulk_against_long_double(lib):
    """4 M samples per family against the long-double library (measured: asin 0.86, atan2 1.6 ulp)."""
    rng = np.random.default_rng(6)
    n = 1_000_000
    for x in (rng.uniform(-1, 1, n), 1 - 10 ** rng.uniform(-16, -0.3, n), 10 ** rng.uniform(-300, -1, n)):
        assert lib.t_err_asin(np.ascontiguousarray(x), n) < 1.0
    sgn = lambda: rng.choice([-1.0, 1.0], n)
    for y, x in ((rng.normal(size=n), rng.normal(size=n)), (rng.uniform(0.40, 0.43, n), np.ones(n)),
                 (10 ** rng.uniform(-20, 20, n) * sgn(), 10 ** rng.uniform(-20, 20, n) * sgn())):
        assert lib.t_err_atan2(np.ascontiguousarray(y), np.ascontiguousarray(x), n) < 1.7


def test_inverse_function_specials(lib):
    """NumPy's results where the trace can reach them: total internal reflection (|x| > 1 -> NaN), dead rays (NaN in ->
    NaN out), a pipe hit on the equator (atan(+-inf) = +-pi/2), axis-aligned and zero arguments with their signs."""
    x = np.array([1.0, -1.0, 0.0, -0.0, 0.5, -0.5])
    assert np.array_equal(_call(lib.t_asin, x), np.arcsin(x)) and np.signbit(_call(lib.t_asin, x)[3])
    assert np.isnan(_call(lib.t_asin, np.array([1.0000000000000002, -2.0, np.nan, np.inf]))).all()
    x = np.array([np.inf, -np.inf, 0.0, -0.0, 1e300, -1e300])
    assert np.array_equal(_call(lib.t_atan, x), np.arctan(x))
    assert np.isnan(_call(lib.t_atan, np.array([np.nan]))).all()
    ys = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 1.0, np.inf, 1.0, 0.0, -0.0, 1e290, 1e-290, 3.0])
    xs = np.array([0.0, 0.0, -0.0, -0.0, 0.0, 0.0, np.inf, 1.0, -np.inf, -1.0, -1.0, 1e-290, 1e290, -3.0])
    got, ref = _atan2(lib, ys, xs), np.arctan2(ys, xs)
    assert np.allclose(got, ref, rtol=3e-16, atol=1e-290) and np.array_equal(np.signbit(got), np.signbit(ref))
    assert np.isnan(_atan2(lib, np.array([np.nan, 1.0]), np.array([1.0, np.nan]))).all()

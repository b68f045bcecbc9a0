"""ctypes bindings to oracle/rt_oracle.c (TEST INFRASTRUCTURE — see oracle/__init__.py)."""
import ctypes as C

import numpy as np

from .build import build

KEYS = ("lens_1_x", "lens_1_z", "pipe_x", "pipe_z", "lens_2_x", "lens_2_z", "target_x", "target_z")

# the reference's constants (main_rt.py:449-457)
REF_LENS = dict(c1=6400.0, c2=1483.0, l0=0.12156646438729327, h0=0.08843353561270673)


class _Lens(C.Structure):
    _fields_ = [("c1", C.c_double), ("c2", C.c_double), ("l0", C.c_double), ("h0", C.c_double),
                ("d", C.c_double)]


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_shoot.argtypes = [C.POINTER(_Lens), C.c_double, C.c_double, C.c_double, C.c_double,
                                _dp, _dp, C.c_int, _dp, _u8p]
        L.orc_shoot.restype = None
        L.orc_tof4.argtypes = [C.POINTER(_Lens), C.c_double, C.c_double, _dp, C.c_int, _dp]
        L.orc_tof4.restype = None
        L.orc_match.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, C.c_double, C.c_double, _u8p, _dp, _i32p]
        L.orc_match.restype = None
        L.orc_ray_hits.argtypes = [_dp, C.c_int, _dp, C.c_int, C.c_double, C.c_double, _u8p]
        L.orc_ray_hits.restype = None
        L.orc_shoot_batch.argtypes = [C.POINTER(_Lens), _dp, C.c_int, _dp, _dp, C.c_int, _dp, _dp,
                                      C.c_int, _dp]
        L.orc_shoot_batch.restype = None
        L.orc_tt_layers.argtypes = [_dp, _dp, C.c_int, _dp, _dp, C.c_int, _dp, _dp, C.c_int, _dp]
        L.orc_tt_layers.restype = None
        L.orc_tt_layers_newton.argtypes = L.orc_tt_layers.argtypes
        L.orc_tt_layers_newton.restype = None
        L.orc_tt_lens.argtypes = [C.POINTER(_Lens), C.c_double, C.c_double, _dp, _dp, C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_tt_lens.restype = None
        L.orc_solve.argtypes = [C.POINTER(_Lens), C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _dp, C.c_int,
                                _dp, C.c_int, C.c_uint, _dp, _dp, _dp]
        L.orc_solve.restype = None
        L.orc_trace_alpha.argtypes = [C.POINTER(_Lens), C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, _dp, _dp, C.c_int, C.c_uint, _dp]
        L.orc_trace_alpha.restype = None
        L.orc_lens_point.argtypes = [C.POINTER(_Lens), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_lens_point.restype = None
        L.orc_num_threads.restype = C.c_int
        L.orc_set_trig_noise.argtypes = [C.c_uint]
        L.orc_set_trig_noise.restype = None
        _lib = L
    return _lib


def _lens(c1, c2, l0, h0, d=None):
    return _Lens(c1, c2, l0, h0, (l0 + h0) if d is None else d)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def shoot(x_a, z_a, z_f, alpha, r_outer, pipe_offset, lens=REF_LENS):
    """One transmit point, one geometry -> (out8 [8,n], status [n])."""
    alpha, z_f = _f64(alpha), _f64(z_f)
    n = alpha.size
    out8 = np.empty((8, n), dtype=np.float64)
    status = np.zeros(n, dtype=np.uint8)
    L = _lens(**lens)
    lib().orc_shoot(C.byref(L), r_outer, pipe_offset, float(x_a), float(z_a), z_f, alpha, n, out8, status)
    return out8, status


def shoot_batch(x_a, z_a, z_f, alpha, geoms, lens=REF_LENS):
    """geoms [n_geom,2] = (r_outer, pipe_offset) -> out8 [n_geom, n_tx, 8, n] (OpenMP)."""
    alpha, z_f, x_a, z_a, geoms = _f64(alpha), _f64(z_f), _f64(x_a), _f64(z_a), _f64(geoms)
    n, n_tx, n_geom = alpha.size, x_a.size, geoms.shape[0]
    out8 = np.empty((n_geom, n_tx, 8, n), dtype=np.float64)
    L = _lens(**lens)
    lib().orc_shoot_batch(C.byref(L), geoms, n_geom, x_a, z_a, n_tx, z_f, alpha, n, out8)
    return out8


def tof4(x_a, z_a, out8, lens=REF_LENS):
    out8 = _f64(out8)
    n = out8.shape[1]
    t = np.empty((4, n), dtype=np.float64)
    L = _lens(**lens)
    lib().orc_tof4(C.byref(L), float(x_a), float(z_a), out8, n, t)
    return t


def match(target_x, tof4_, x_rx, atol, rtol=1e-5):
    target_x, tof4_, x_rx = _f64(target_x), _f64(tof4_), _f64(x_rx)
    n, n_rx = target_x.size, x_rx.size
    hit = np.zeros(n_rx, dtype=np.uint8)
    tof = np.zeros(n_rx, dtype=np.float64)
    first = np.zeros(n_rx, dtype=np.int32)
    lib().orc_match(target_x, tof4_, n, x_rx, n_rx, atol, rtol, hit, tof, first)
    return hit.astype(bool), tof, first


def ray_hits(target_x, x_rx, atol, rtol=1e-5):
    target_x, x_rx = _f64(target_x), _f64(x_rx)
    hit = np.zeros(target_x.size, dtype=np.uint8)
    lib().orc_ray_hits(target_x, target_x.size, x_rx, x_rx.size, atol, rtol, hit)
    return hit.astype(bool)


def tt_layers(z_if, c, xe, ze, xf, zf):
    """Planar-layer Fermat travel times [n_e, n_f] by long-double bisection (parity unpinned w.r.t. the reference,
    which has no planar interfaces; pinned bit for bit to 50-digit values: tests/golden/planar_mp.npz)."""
    z_if, c, xe, ze, xf, zf = map(_f64, (z_if, c, xe, ze, xf, zf))
    tt = np.empty((xe.size, xf.size), dtype=np.float64)
    lib().orc_tt_layers(z_if, c, z_if.size, xe, ze, xe.size, xf, zf, xf.size, tt)
    return tt


def tt_layers_newton(z_if, c, xe, ze, xf, zf):
    """fp64 Newton CPU port of the same solve (bench.py's cpu_baseline leg; OpenMP)."""
    z_if, c, xe, ze, xf, zf = map(_f64, (z_if, c, xe, ze, xf, zf))
    tt = np.empty((xe.size, xf.size), dtype=np.float64)
    lib().orc_tt_layers_newton(z_if, c, z_if.size, xe, ze, xe.size, xf, zf, xf.size, tt)
    return tt


def tt_lens(xe, ze, xf, zf, a_lo, a_hi, lens=REF_LENS):
    """Two-point Fermat times through the lens surface by golden-section search (long double)."""
    xe, ze, xf, zf = map(_f64, (xe, ze, xf, zf))
    tt = np.empty((xe.size, xf.size), dtype=np.float64)
    al = np.empty((xe.size, xf.size), dtype=np.float64)
    L = _lens(**lens)
    lib().orc_tt_lens(C.byref(L), a_lo, a_hi, xe, ze, xe.size, xf, zf, xf.size, tt, al)
    return tt, al


TRUE_TANGENT, ANALYTIC_LENS = 0x2, 0x4


def solve(x_a, z_a, z_land, alpha, x_rx, r_outer, pipe_offset, flags=0, lens=REF_LENS):
    """Root-finding pulse-echo times for one geometry / one tx ->
    (tt_min [n_rx], tt_all [n_rx, 4], alpha_all [n_rx, 4]); roots in ascending alpha, NaN padded."""
    alpha, x_rx = _f64(alpha), _f64(x_rx)
    tt = np.empty(x_rx.size, dtype=np.float64)
    ta = np.empty((x_rx.size, 4), dtype=np.float64)
    aa = np.empty((x_rx.size, 4), dtype=np.float64)
    L = _lens(**lens)
    lib().orc_solve(C.byref(L), r_outer, pipe_offset, float(x_a), float(z_a), float(z_land), alpha, alpha.size,
                    x_rx, x_rx.size, flags, tt, ta, aa)
    return tt, ta, aa


def trace_alpha(x_a, z_a, z_f, a, alpha_grid, r_outer, pipe_offset, flags=0, lens=REF_LENS):
    """One ray at an arbitrary launch angle against the polyline of alpha_grid -> out8 [8]."""
    alpha_grid = _f64(alpha_grid)
    L = _lens(**lens)
    xc = np.empty(alpha_grid.size); zc = np.empty(alpha_grid.size)
    x, z = C.c_double(), C.c_double()
    for i, al in enumerate(alpha_grid):
        lib().orc_lens_point(C.byref(L), float(al), C.byref(x), C.byref(z))
        xc[i], zc[i] = x.value, z.value
    o = np.empty(8)
    lib().orc_trace_alpha(C.byref(L), r_outer, pipe_offset, float(x_a), float(z_a), float(z_f), float(a), xc, zc,
                          alpha_grid.size, flags, o)
    return o


def set_trig_noise(mode):
    """Sensitivity probe of the forward trace (see rt_oracle.c): two bits per trigonometric call site move its result by
    0 / +1 / -1 / +2 ulp.  0 = plain libm — ALWAYS reset it after use."""
    lib().orc_set_trig_noise(int(mode) & 0xffffffff)


def num_threads():
    return int(lib().orc_num_threads())

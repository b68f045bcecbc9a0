"""Host-side mirror of the reference's call surface, on top of the C ABI (include/rtus.h).

``shoot_rays(x_a, z_a, z_f, alpha, plot=False)`` keeps the positional signature and the 8-key
dict of float64[N] of the reference (main_rt.py:337, 432-441).  The constants the reference
reads as module globals (c1, c2, l0, h0, d, r_outer, pipe_offset — main_rt.py:449-467) are an
explicit :class:`Params`; when none is given they are looked up the way the reference's scripts
define them (``configure(...)`` first, then same-named attributes of ``__main__``).

Everything here runs on the GPU through librtus.so.  There is no CPU fallback.
"""
import ctypes as C
import sys
import warnings
from dataclasses import dataclass, replace

import numpy as np

from . import _lib
from ._lib import Lens

KEYS = ("lens_1_x", "lens_1_z", "pipe_x", "pipe_z", "lens_2_x", "lens_2_z", "target_x", "target_z")


@dataclass(frozen=True)
class Params:
    """Medium + geometry constants (reference: module globals, main_rt.py:449-467)."""
    c1: float = 6400.0                     # main_rt.py:449
    c2: float = 1483.0                     # main_rt.py:450
    l0: float = 0.12156646438729327        # main_rt.py:453
    h0: float = 0.08843353561270673        # main_rt.py:454
    d: float = None                        # main_rt.py:455 (l0 + h0 when None)
    r_outer: float = 0.05                  # main_rt.py:466
    pipe_offset: float = 0.0               # main_rt.py:467

    def __post_init__(self):
        if self.d is None:
            object.__setattr__(self, "d", float(np.float64(self.l0) + np.float64(self.h0)))

    def lens(self) -> Lens:
        return Lens(float(self.c1), float(self.c2), float(self.l0), float(self.h0), float(self.d))


#: the reference's launch-angle half-aperture (main_rt.py:457)
ALPHA_MAX = float(np.float64(50.62033040986099 * (np.pi / 180)))

_configured = None


def configure(params: Params = None, **kw) -> Params:
    """Set the default Params (the explicit replacement for assigning module globals)."""
    global _configured
    base = params if params is not None else (_configured or Params())
    _configured = replace(base, **kw) if kw else base
    return _configured


def _resolve(params):
    if params is not None:
        return params
    if _configured is not None:
        return _configured
    main = sys.modules.get("__main__")
    names = ("c1", "c2", "l0", "h0", "d", "r_outer", "pipe_offset")
    if main is not None and all(hasattr(main, n) for n in names):   # main_compare.py-style script
        return Params(**{n: float(getattr(main, n)) for n in names})
    raise ValueError("no Params given: pass params=..., call configure(...), or define "
                     "c1,c2,l0,h0,d,r_outer,pipe_offset in __main__ as the reference scripts do")


def reference_elements(num_elements=64, pitch=0.0006):
    """main_rt.py:469-474 — centred linear array plus the virtual centre element at index 32."""
    x_a = np.arange(num_elements, dtype=np.float64) * np.float64(pitch)
    x_a = x_a - np.mean(x_a)
    return np.insert(x_a, num_elements // 2, np.float64(0.0))


def _f64(a, name, ndim=1):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim == 1:
        a = np.atleast_1d(a)
    if a.ndim != ndim:
        raise ValueError(f"{name} must be {ndim}-D, got shape {a.shape}")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data


def _out(out, shape, dtype, name="out"):
    """The caller's result buffer (checked) or a fresh pageable one."""
    if out is None:
        return np.empty(shape, dtype=dtype)
    if not isinstance(out, np.ndarray) or out.dtype != np.dtype(dtype) or out.shape != tuple(shape) \
            or not out.flags.c_contiguous or not out.flags.writeable:
        raise ValueError(f"{name} must be a writeable C-contiguous {np.dtype(dtype).name} array of shape {tuple(shape)}")
    return out


SHOOT_FAST_MATH = 0x1       # RTUS_SHOOT_FAST_MATH
TRUE_PIPE_TANGENT = 0x2     # RTUS_TRUE_PIPE_TANGENT  (physically correct; not the reference)
ANALYTIC_LENS = 0x4         # RTUS_ANALYTIC_LENS      (physically correct; not the reference)
MAX_ROOTS = 4               # RTUS_MAX_ROOTS


def _flags(fast=False, true_tangent=False, analytic_lens=False):
    return (SHOOT_FAST_MATH if fast else 0) | (TRUE_PIPE_TANGENT if true_tangent else 0) | \
        (ANALYTIC_LENS if analytic_lens else 0)


def shoot_batch(x_a, z_a, z_f, alpha, geoms=None, *, params: Params = None, want=("out8",), fast=False,
                true_tangent=False, analytic_lens=False, device=0):
    """Forward trace for n_geom geometries x n_tx transmit points in ONE launch.

    geoms: [n_geom, 2] of (r_outer, pipe_offset); default = the one geometry in ``params``.
    want:  any of "out8" [G,T,8,N], "tof4" [G,T,4,N], "tof" [G,T,N], "land_x" [G,T,N], "status".
    fast:  vector-form arithmetic (no trigonometry; measured 1.6-1.7x faster); the default reproduces the reference's
           angle-form arithmetic operation for operation.
    """
    p = _resolve(params)
    x_a, z_a = _f64(x_a, "x_a"), _f64(z_a, "z_a")
    alpha, z_f = _f64(alpha, "alpha"), _f64(z_f, "z_f")
    if x_a.shape != z_a.shape:
        raise ValueError("x_a and z_a must have the same length")        # main_rt.py:28-29
    if alpha.shape != z_f.shape:
        raise ValueError("alpha and z_f must have the same length")
    if alpha.size < 2:
        raise ValueError("Curve needs at least two points.")             # main_rt.py:26-27
    geoms = (np.asarray([[p.r_outer, p.pipe_offset]], dtype=np.float64) if geoms is None
             else _f64(geoms, "geoms", 2))
    if geoms.shape[1] != 2:
        raise ValueError("geoms must be [n_geom, 2] = (r_outer, pipe_offset)")
    G, T, N = geoms.shape[0], x_a.size, alpha.size
    bufs = dict(out8=None, tof4=None, tof=None, land_x=None, status=None)
    shapes = dict(out8=(G, T, 8, N), tof4=(G, T, 4, N), tof=(G, T, N), land_x=(G, T, N), status=(G, T, N))
    for w in want:
        if w not in bufs:
            raise ValueError(f"unknown output {w!r}")
        bufs[w] = np.empty(shapes[w], dtype=np.uint8 if w == "status" else np.float64)
    lens = p.lens()
    st = _lib.lib().rtus_shoot(C.byref(lens), _ptr(geoms), G, _ptr(x_a), _ptr(z_a), T, _ptr(alpha), _ptr(z_f), N,
                               _ptr(bufs["out8"]), _ptr(bufs["tof4"]), _ptr(bufs["tof"]), _ptr(bufs["land_x"]),
                               _ptr(bufs["status"]), _flags(fast, true_tangent, analytic_lens), int(device))
    _lib.check(st, "rtus_shoot")
    return {w: bufs[w] for w in want}


def shoot_rays(x_a, z_a, z_f, alpha, plot=False, *, params: Params = None, device=0, **options):
    """Drop-in for the reference's shoot_rays (main_rt.py:337): one transmit point, one geometry.

    Returns the same dict of eight float64[N] arrays (main_rt.py:432-441); invalid rays are NaN.
    ``plot`` is accepted for signature compatibility; plotting (main_rt.py:407-430) is out of scope
    and the default is False so the call never blocks on a GUI.  ``options``: fast / true_tangent /
    analytic_lens as in :func:`shoot_batch` (all off = the reference's arithmetic).

    The reference's driver calls this 210 times in a row (main_rt.py:464-482), so the wrapper itself is kept
    short: one result buffer whose eight rows are the eight arrays, no per-key copies.
    """
    if plot:
        warnings.warn("rtus.shoot_rays: plotting is not part of the accelerated path; ignoring plot=True",
                      stacklevel=2)
    if np.ndim(x_a) != 0 or np.ndim(z_a) != 0:
        raise ValueError("x_a and z_a are scalars (one transmit point), as in main_rt.py:482")
    p = _resolve(params)
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    z_f = np.ascontiguousarray(z_f, dtype=np.float64)
    if alpha.ndim != 1 or z_f.ndim != 1:
        raise ValueError("alpha and z_f must be 1-D")
    if alpha.shape != z_f.shape:
        raise ValueError("alpha and z_f must have the same length")
    if alpha.size < 2:
        raise ValueError("Curve needs at least two points.")             # main_rt.py:26-27
    out8 = np.empty((8, alpha.size), dtype=np.float64)
    head = np.array([p.r_outer, p.pipe_offset, x_a, z_a], dtype=np.float64)   # geoms[1][2], x_a[1], z_a[1]
    hp = head.ctypes.data
    st = _lib.lib().rtus_shoot(C.byref(p.lens()), hp, 1, hp + 16, hp + 24, 1, alpha.ctypes.data, z_f.ctypes.data, alpha.size,
                               out8.ctypes.data, None, None, None, None, _flags(**options), int(device))
    _lib.check(st, "rtus_shoot")
    return dict(zip(KEYS, out8))          # eight row views of one caller-owned buffer


def match_elements(land_x, tof, x_rx, atol=1e-6, rtol=1e-5, *, device=0):
    """Element matcher (main_rt.py:487-501): per receive element the first ray within np.isclose.

    land_x, tof: [..., N]; x_rx: [E].  Returns (hit bool[..., E], tof_hit f64[..., E] (0.0 if none),
    first_ray int32[..., E] (-1 if none)).
    """
    land_x = np.ascontiguousarray(land_x, dtype=np.float64)
    tof = np.ascontiguousarray(tof, dtype=np.float64)
    if land_x.shape != tof.shape or land_x.ndim < 1:
        raise ValueError("land_x and tof must have the same shape [..., n_rays]")
    x_rx = _f64(x_rx, "x_rx")
    lead, N, E = land_x.shape[:-1], land_x.shape[-1], x_rx.size
    nb = int(np.prod(lead)) if lead else 1
    first = np.empty((nb, E), dtype=np.int32)
    hit = np.empty((nb, E), dtype=np.uint8)
    tof_hit = np.empty((nb, E), dtype=np.float64)
    st = _lib.lib().rtus_match(_ptr(land_x), _ptr(tof), nb, N, _ptr(x_rx), E, float(atol), float(rtol),
                               _ptr(first), _ptr(hit), _ptr(tof_hit), int(device))
    _lib.check(st, "rtus_match")
    return (hit.astype(bool).reshape(lead + (E,)), tof_hit.reshape(lead + (E,)), first.reshape(lead + (E,)))


def sweep_batch(x_a, z_a, z_f, alpha, x_rx, geoms=None, *, atol=1e-6, rtol=1e-5, params: Params = None, want=(), fast=False,
                true_tangent=False, analytic_lens=False, device=0):
    """One body of the reference's parameter loop (main_rt.py:464-501) for n_geom geometries x n_tx transmit points in ONE kernel:
    ``shoot_batch`` and ``match_elements`` fused — the matcher runs on the landing points while they are in registers.

    Returns {"hit" bool[G,T,E], "tof_hit" f64[G,T,E] (0.0 where no ray hits), "first_ray" int32[G,T,E] (-1 where none)} plus the
    per-ray arrays named in ``want`` ("tof", "land_x": [G,T,N]).  Bit-identical to the two calls it replaces.
    """
    p = _resolve(params)
    x_a, z_a = _f64(x_a, "x_a"), _f64(z_a, "z_a")
    alpha, z_f, x_rx = _f64(alpha, "alpha"), _f64(z_f, "z_f"), _f64(x_rx, "x_rx")
    if x_a.shape != z_a.shape:
        raise ValueError("x_a and z_a must have the same length")        # main_rt.py:28-29
    if alpha.shape != z_f.shape:
        raise ValueError("alpha and z_f must have the same length")
    if alpha.size < 2:
        raise ValueError("Curve needs at least two points.")             # main_rt.py:26-27
    geoms = (np.asarray([[p.r_outer, p.pipe_offset]], dtype=np.float64) if geoms is None
             else _f64(geoms, "geoms", 2))
    if geoms.shape[1] != 2:
        raise ValueError("geoms must be [n_geom, 2] = (r_outer, pipe_offset)")
    G, T, N, E = geoms.shape[0], x_a.size, alpha.size, x_rx.size
    bufs = dict(tof=None, land_x=None)
    for w in want:
        if w not in bufs:
            raise ValueError(f"unknown output {w!r}")
        bufs[w] = np.empty((G, T, N), dtype=np.float64)
    first = np.empty((G, T, E), dtype=np.int32)
    hit = np.empty((G, T, E), dtype=np.uint8)
    tof_hit = np.empty((G, T, E), dtype=np.float64)
    lens = p.lens()
    st = _lib.lib().rtus_sweep(C.byref(lens), _ptr(geoms), G, _ptr(x_a), _ptr(z_a), T, _ptr(alpha), _ptr(z_f), N, _ptr(x_rx), E,
                               float(atol), float(rtol), _ptr(first), _ptr(hit), _ptr(tof_hit), _ptr(bufs["tof"]), _ptr(bufs["land_x"]),
                               _flags(fast, true_tangent, analytic_lens), int(device))
    _lib.check(st, "rtus_sweep")
    out = {"hit": hit.astype(bool), "tof_hit": tof_hit, "first_ray": first}
    out.update({w: bufs[w] for w in want})
    return out


def ray_hits(land_x, x_rx, atol=1e-4, rtol=1e-5, *, device=0):
    """Per ray: does any element match (main_compare.py:518-521)."""
    land_x = np.ascontiguousarray(land_x, dtype=np.float64)
    x_rx = _f64(x_rx, "x_rx")
    lead, N = land_x.shape[:-1], land_x.shape[-1]
    nb = int(np.prod(lead)) if lead else 1
    rh = np.empty((nb, N), dtype=np.uint8)
    st = _lib.lib().rtus_ray_hits(_ptr(land_x), nb, N, _ptr(x_rx), x_rx.size, float(atol), float(rtol),
                                  _ptr(rh), int(device))
    _lib.check(st, "rtus_ray_hits")
    return rh.astype(bool).reshape(lead + (N,))


def _device_list(devices):
    d = np.ascontiguousarray(devices, dtype=np.int32).reshape(-1)
    if d.size == 0:
        raise ValueError("devices must list at least one GPU")
    return d


TAUP_TAIL = 0x1             # RTUS_TT_TAUP_TAIL (include/rtus.h)


def travel_time_layers(z_if, c, xe, ze, xf, zf, *, return_iters=False, out=None, device=0, devices=None, taup=False):
    """Element x focal-point Fermat travel times through horizontal layers -> tt[n_e, n_f].

    ``out``: optional float64 [n_e, n_f] result buffer, returned as ``tt``.
    ``devices``: a list of GPU indices — the table's rows are solved in contiguous blocks on all of them at once and each
    GPU copies its block straight into ``tt`` (rtus_tt_layers_multi_ex); the result is bit for bit the one-GPU table.
    ``taup``: the faster accuracy tier (RTUS_TT_TAUP_TAIL: the tau-p form of the travel time, <= 6e-11 relative at worst,
    measured 3e-17 s on BASELINE configs[2]; the default tier: <= 1e-13 relative) — the tier bench.py's headline times.
    The aperture may come in any order: the rows are solved in (depth, position) order and stored where they belong, so an
    element's bits do not depend on it (``return_iters`` takes the elements as given: it is a diagnostic of that).

    NOT in the reference (no planar interfaces there): parity unpinned, see DESIGN.md.
    """
    z_if = _f64(z_if, "z_if") if np.size(z_if) else np.zeros(0)
    c = _f64(c, "c")
    if c.size != z_if.size + 1:
        raise ValueError("need len(c) == len(z_if) + 1")
    xe, ze, xf, zf = _f64(xe, "xe"), _f64(ze, "ze"), _f64(xf, "xf"), _f64(zf, "zf")
    if xe.shape != ze.shape or xf.shape != zf.shape:
        raise ValueError("xe/ze and xf/zf must pair up")
    if taup and return_iters:
        raise ValueError("return_iters is a diagnostic of the default tier")
    flags = TAUP_TAIL if taup else 0
    tt = _out(out, (xe.size, xf.size), np.float64)
    if devices is not None:
        if return_iters:
            raise ValueError("return_iters is a one-device diagnostic")
        dv = _device_list(devices)
        st = _lib.lib().rtus_tt_layers_multi_ex(_ptr(z_if) if z_if.size else None, _ptr(c), z_if.size, _ptr(xe), _ptr(ze), xe.size,
                                                _ptr(xf), _ptr(zf), xf.size, _ptr(tt), dv.ctypes.data_as(C.POINTER(C.c_int)), dv.size, flags)
        _lib.check(st, "rtus_tt_layers_multi_ex")
        return tt
    iters = np.empty((xe.size, xf.size), dtype=np.uint8) if return_iters else None
    st = _lib.lib().rtus_tt_layers_ex(_ptr(z_if) if z_if.size else None, _ptr(c), z_if.size, _ptr(xe), _ptr(ze),
                                      xe.size, _ptr(xf), _ptr(zf), xf.size, _ptr(tt), _ptr(iters), flags, int(device))
    _lib.check(st, "rtus_tt_layers_ex")
    return (tt, iters) if return_iters else tt


def travel_time_lens(xe, ze, xf, zf, *, params: Params = None, alpha_lo=None, alpha_hi=None, dtype=np.float64,
                     return_alpha=False, out=None, device=0, devices=None):
    """Element x focal-point Fermat travel times through the reference's curved lens surface
    (h(alpha) of main_rt.py:180-189): elements in the lens (c1), targets in the water (c2) -> tt[n_e, n_f].

    dtype float32 runs the fp32 kernel (BASELINE config 4).  ``return_alpha`` also returns the polar
    angle of the refraction point.  As a two-point solver this is not in the reference; it is pinned to
    it through Fermat <=> Snell (see include/rtus.h).
    """
    p = _resolve(params)
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise ValueError("dtype must be float64 or float32")
    a_lo = -ALPHA_MAX if alpha_lo is None else float(alpha_lo)
    a_hi = ALPHA_MAX if alpha_hi is None else float(alpha_hi)
    arr = [np.atleast_1d(np.ascontiguousarray(v, dtype=dt)) for v in (xe, ze, xf, zf)]
    xe, ze, xf, zf = arr
    if xe.shape != ze.shape or xf.shape != zf.shape or xe.ndim != 1 or xf.ndim != 1:
        raise ValueError("xe/ze and xf/zf must be 1-D and pair up")
    tt = _out(out, (xe.size, xf.size), dt)                       # out: optional caller-owned result buffer
    al = np.empty((xe.size, xf.size), dtype=dt) if return_alpha else None
    lens = p.lens()
    if devices is not None:                      # fp32 table over several GPUs (BASELINE configs[3]): rtus_tt_lens_f32_multi
        if dt != np.float32 or return_alpha:
            raise ValueError("devices=[...] is the float32 table without the alpha output")
        dv = _device_list(devices)
        st = _lib.lib().rtus_tt_lens_f32_multi(C.byref(lens), a_lo, a_hi, _ptr(xe), _ptr(ze), xe.size, _ptr(xf), _ptr(zf), xf.size,
                                               _ptr(tt), dv.ctypes.data_as(C.POINTER(C.c_int)), dv.size)
        _lib.check(st, "rtus_tt_lens_f32_multi")
        return tt
    fn = _lib.lib().rtus_tt_lens if dt == np.float64 else _lib.lib().rtus_tt_lens_f32
    st = fn(C.byref(lens), a_lo, a_hi, _ptr(xe), _ptr(ze), xe.size, _ptr(xf), _ptr(zf), xf.size, _ptr(tt), _ptr(al),
            int(device))
    _lib.check(st, "rtus_tt_lens")
    return (tt, al) if return_alpha else tt


def fmc_table_layers(z_if, c, x_tx, x_rx, z_reflector, *, z_array=0.0, device=0, devices=None, taup=False):
    """Full-matrix-capture tx/rx travel-time table for a planar specular reflector at depth z_reflector
    under horizontal layers (BASELINE config 5).  The down-and-up path through the layers is unfolded
    about the reflector plane into a one-way path through the mirrored stack, so the table is one
    travel_time_layers call: tt[n_tx, n_rx].  Not in the reference (parity unpinned)."""
    z_if = np.asarray(z_if, dtype=np.float64).reshape(-1)
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    if c.size != z_if.size + 1:
        raise ValueError("need len(c) == len(z_if) + 1")
    above = z_if < z_reflector
    zi, cc = z_if[above], c[:above.sum() + 1]
    z_m = np.concatenate([zi, (2.0 * z_reflector - zi)[::-1]])          # mirrored interfaces
    c_m = np.concatenate([cc, cc[::-1][1:]])                            # ... and speeds (reflector layer merged)
    x_tx, x_rx = _f64(x_tx, "x_tx"), _f64(x_rx, "x_rx")
    return travel_time_layers(z_m, c_m, x_tx, np.full(x_tx.size, float(z_array)), x_rx,
                              np.full(x_rx.size, 2.0 * z_reflector - float(z_array)), device=device, devices=devices, taup=taup)


SOLVE_ONE_LANE = 0x10       # RTUS_SOLVE_ONE_LANE (include/rtus.h)
SOLVE_THREE_LAUNCHES = 0x20


def solve_travel_times(x_a, z_a, x_rx, alpha, geoms=None, *, z_land=None, params: Params = None, fast=False,
                       true_tangent=False, analytic_lens=False, all_roots=False, one_lane=False, three_launches=False, device=0):
    """Pulse-echo travel times tx -> lens -> pipe -> lens -> rx by root-finding x_land(alpha) = x_rx — the
    replacement for the reference's grid scan + tolerance matcher (main_rt.py:479-501).

    Returns tt [G, T, E] (least time over the element's ray paths, NaN if none) and the launch angle
    alpha_root [G, T, E]; with all_roots also (tt_all, alpha_all) [G, T, E, 4] in ascending alpha and
    n_roots [G, T, E].  ``true_tangent`` / ``analytic_lens`` switch on the physically-correct variants.
    ``one_lane``: refine every bracket by one lane whatever the size of the call (RTUS_SOLVE_ONE_LANE; calls of up to 32,768
    (row, element) pairs otherwise use three lanes per bracket — same tolerances, other last bits).
    """
    p = _resolve(params)
    x_a, z_a = _f64(x_a, "x_a"), _f64(z_a, "z_a")
    alpha, x_rx = _f64(alpha, "alpha"), _f64(x_rx, "x_rx")
    if x_a.shape != z_a.shape:
        raise ValueError("x_a and z_a must have the same length")
    if alpha.size < 2:
        raise ValueError("Curve needs at least two points.")
    if not np.all(np.diff(alpha) > 0):
        raise ValueError("alpha must be strictly ascending (its intervals are the root brackets)")
    geoms = (np.asarray([[p.r_outer, p.pipe_offset]], dtype=np.float64) if geoms is None
             else _f64(geoms, "geoms", 2))
    G, T, E = geoms.shape[0], x_a.size, x_rx.size
    z_land = p.d if z_land is None else float(z_land)
    tt = np.empty((G, T, E)); ar = np.empty((G, T, E))
    ta = np.empty((G, T, E, MAX_ROOTS)) if all_roots else None
    aa = np.empty((G, T, E, MAX_ROOTS)) if all_roots else None
    nr = np.empty((G, T, E), dtype=np.uint8) if all_roots else None
    lens = p.lens()
    st = _lib.lib().rtus_solve(C.byref(lens), _ptr(geoms), G, _ptr(x_a), _ptr(z_a), T, _ptr(alpha), alpha.size,
                               _ptr(x_rx), E, z_land, _ptr(tt), _ptr(ar), _ptr(ta), _ptr(aa), _ptr(nr),
                               _flags(fast, true_tangent, analytic_lens) | (SOLVE_ONE_LANE if one_lane else 0) | (SOLVE_THREE_LAUNCHES if three_launches else 0),
                               int(device))
    _lib.check(st, "rtus_solve")
    return (tt, ar, ta, aa, nr) if all_roots else (tt, ar)


def focal_delays(tt, *, out=None, device=0):
    """Transmit focal law from a travel-time table tt[n_elem, n_focal]: the delay each element must be fired with so that
    all wavefronts reach the focal point together, delays[e, f] = max_e' tt[e', f] - tt[e, f] (elements without a ray
    path — NaN — are ignored by the maximum and stay NaN).  SURVEY 8(f) row 4; rtus_focal_delays on the GPU."""
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    if tt.ndim != 2:
        raise ValueError("tt must be [n_elem, n_focal]")
    d = _out(out, tt.shape, np.float64)
    st = _lib.lib().rtus_focal_delays(_ptr(tt), tt.shape[0], tt.shape[1], _ptr(d), int(device))
    _lib.check(st, "rtus_focal_delays")
    return d


def tfm_image(fmc, fs, tt_tx, tt_rx=None, *, t0=0.0, out=None, device=0):
    """Total-focusing-method delay-and-sum over full-matrix-capture data: image[f] = sum over (tx, rx) of the A-scan
    fmc[tx, rx, :] (float32, ``fs`` samples per second, first sample at time ``t0``) linearly interpolated at
    tt_tx[tx, f] + tt_rx[rx, f]; tt_rx defaults to tt_tx (same aperture transmits and receives).  The travel-time tables
    are what travel_time_layers / travel_time_lens return.  Pairs without a ray path (NaN) contribute nothing; samples
    outside a record count as zero.  -> float32 [n_focal].  SURVEY 8(f) row 4; not in the reference."""
    fmc = np.ascontiguousarray(fmc, dtype=np.float32)
    if fmc.ndim != 3:
        raise ValueError("fmc must be [n_tx, n_rx, n_t]")
    tt_tx = np.ascontiguousarray(tt_tx, dtype=np.float64)
    same = tt_rx is None or tt_rx is tt_tx
    tt_rx = tt_tx if same else np.ascontiguousarray(tt_rx, dtype=np.float64)
    if tt_tx.ndim != 2 or tt_rx.ndim != 2 or tt_tx.shape[1] != tt_rx.shape[1]:
        raise ValueError("tt_tx / tt_rx must be [n_tx, n_focal] / [n_rx, n_focal]")
    if tt_tx.shape[0] != fmc.shape[0] or tt_rx.shape[0] != fmc.shape[1]:
        raise ValueError("fmc's first two dimensions must match the rows of tt_tx and tt_rx")
    img = _out(out, (tt_tx.shape[1],), np.float32)
    st = _lib.lib().rtus_tfm(_ptr(fmc), fmc.shape[0], fmc.shape[1], fmc.shape[2], float(fs), float(t0), _ptr(tt_tx), _ptr(tt_rx),
                             tt_tx.shape[1], _ptr(img), int(device))
    _lib.check(st, "rtus_tfm")
    return img

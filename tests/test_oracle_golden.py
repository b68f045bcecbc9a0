"""CPU: pin the oracle (oracle/rt_oracle.c) to the reference.

Golden vectors = outputs of the reference itself (tests/golden/make_golden.py ran
main_rt.shoot_rays in the build container) + the reference's own compare.csv / database_2.csv.
Tolerances: positions 1e-12 m, times 1e-15 s (observed: ~1e-14 m, ~1e-18 s — libm ulps).
"""
import csv
import os

import numpy as np

from conftest import D_PLANE, GOLDEN, load_golden, max_abs, nan_equal_mask
from oracle import cport

POS_TOL, TIME_TOL = 1e-12, 1e-15


def _check8(o, r, tag):
    for k in range(8):
        assert nan_equal_mask(o[k], r[k]), f"{tag}: NaN mask differs in slot {k}"
        assert max_abs(o[k], r[k]) < POS_TOL, f"{tag}: slot {k}"


def test_oracle_vs_reference_compare_config():
    g = load_golden("compare_cfg.npz")
    o, st = cport.shoot(0.0, D_PLANE, g["zf"], g["alpha"], 0.037, 0.0038)
    _check8(o, g["out8"], "compare")
    t4 = cport.tof4(0.0, D_PLANE, o)
    assert nan_equal_mask(t4, g["tof4"]) and max_abs(t4, g["tof4"]) < TIME_TOL
    assert st.sum() == 0


def test_oracle_vs_compare_csv():
    """The reference's own output file (main_compare.py:502-524): alpha, 4 segment times, hitted."""
    rows = list(csv.reader(open(os.path.join(GOLDEN, "compare.csv"))))
    assert rows[0] == ["alpha", "offset", "radius", "hitted", "tof_1", "tof_2", "tof_3", "tof_4"]
    rows = rows[1:]
    assert len(rows) == 1810
    alpha_csv = np.array([float(r[0]) for r in rows])
    alpha_max = np.float64(50.62033040986099 * (np.pi / 180))
    alpha = np.linspace(-alpha_max, alpha_max, 1810)
    assert np.array_equal(alpha, alpha_csv)                       # R11: the grid is bit-exact
    tofs = np.array([[float(v) for v in r[4:8]] for r in rows]).T
    hitted = np.array([r[3] == "True" for r in rows])
    o, _ = cport.shoot(0.0, D_PLANE, np.full(1810, D_PLANE), alpha, 0.037, 0.0038)
    t4 = cport.tof4(0.0, D_PLANE, o)
    assert nan_equal_mask(t4, tofs)
    assert max_abs(t4, tofs) < TIME_TOL
    x_elem = load_golden("compare_cfg.npz")["x_elem"]
    assert np.array_equal(cport.ray_hits(o[6], x_elem, 1e-4), hitted)    # main_compare.py:518-521
    assert hitted.sum() == 393


def test_oracle_vs_database2_csv():
    """The reference's sweep output (main_rt.py:464-504): 10 radii x 21 offsets x 65 elements."""
    rows = list(csv.reader(open(os.path.join(GOLDEN, "database_2.csv"))))[1:]
    assert len(rows) == 13650
    db_hit = np.array([r[3] == "True" for r in rows]).reshape(210, 65)
    db_tof = np.array([float(r[4]) for r in rows]).reshape(210, 65)
    s = load_golden("sweep_cfg.npz")
    alpha, geoms, x_elem = s["alpha"], s["geoms"], s["x_elem"]
    zf = np.full(905, D_PLANE)
    out = cport.shoot_batch([0.0], [D_PLANE], zf, alpha, geoms)
    for gi in range(210):
        o = out[gi, 0]
        t4 = cport.tof4(0.0, D_PLANE, o)
        hit, tof, _ = cport.match(o[6], t4, x_elem, 1e-6)
        assert np.array_equal(hit, db_hit[gi]), f"geometry {gi}"
        assert np.max(np.abs(tof - db_tof[gi])) < TIME_TOL
        ref = s["target_x_tof"][gi]
        assert nan_equal_mask(o[6], ref[0]) and max_abs(o[6], ref[0]) < 1e-11
    assert db_hit.sum() == 498


def test_oracle_vs_reference_edges_alltx_nesting():
    e = load_golden("edge_cfg.npz")
    for tag in ("q1nan", "tir", "off0", "offtx"):
        r_o, off, x_tx = e[tag + "_cfg"]
        o, _ = cport.shoot(x_tx, D_PLANE, np.full(905, D_PLANE), e["alpha"], r_o, off)
        _check8(o, e[tag], tag)
    assert int(e["miss_raises"]) == 1            # the reference raises on a missed pipe
    o, st = cport.shoot(0.0, D_PLANE, np.full(905, D_PLANE), e["alpha"], 0.005, 0.05)
    assert st.any() and np.isnan(o[2][st.astype(bool)]).all() and np.isnan(o[6][st.astype(bool)]).all()
    for tag in ("a", "b"):
        g = load_golden(f"alltx_{tag}.npz")
        for t in range(0, 65, 4):
            o, _ = cport.shoot(g["x_elem"][t], D_PLANE, np.full(905, D_PLANE), g["alpha"],
                               float(g["r_outer"]), float(g["pipe_offset"]))
            _check8(o, g["out8"][t], f"alltx_{tag}[{t}]")
    n = load_golden("nest_cfg.npz")
    alpha_max = np.float64(50.62033040986099 * (np.pi / 180))
    for k in (181, 1809, 3617):
        o, _ = cport.shoot(0.0, D_PLANE, np.full(k, D_PLANE), np.linspace(-alpha_max, alpha_max, k), 0.037, 0.0038)
        _check8(o, n[f"n{k}"], f"n{k}")


def test_line_curve_corner_cases_vs_reference():
    """find_line_curve_intersection semantics (main_rt.py:78-168): hand-made polylines run through the
    REFERENCE's own function by tests/golden/make_golden.py (line_curve_cases.npz); the oracle must agree."""
    import ctypes as C
    L = cport.lib()
    L.orc_line_curve.argtypes = [C.c_double, C.c_double, cport._dp, cport._dp, C.c_int,
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_line_curve.restype = C.c_int
    g = load_golden("line_curve_cases.npz")
    tags = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    assert len(tags) >= 9
    for tag in tags:
        v = g[tag + "_in"]
        m, b, n = v[0], v[1], (v.size - 2) // 2
        xc, yc = np.ascontiguousarray(v[2:2 + n]), np.ascontiguousarray(v[2 + n:])
        xi, yi = C.c_double(), C.c_double()
        ok = L.orc_line_curve(m, b, xc, yc, n, C.byref(xi), C.byref(yi))
        exp = g[tag + "_out"]
        if np.isnan(exp[0]):
            assert not ok, tag                                 # the reference returned (None, None)
        else:
            assert ok, tag
            assert abs(xi.value - exp[0]) < 1e-12 and abs(yi.value - exp[1]) < 1e-9 * max(1.0, abs(exp[1])), tag
    # NaN line -> sign change "found" at 0, then the bounds check fails -> None (np.sign(NaN) semantics)
    xi, yi = C.c_double(), C.c_double()
    assert not L.orc_line_curve(float("nan"), 0.0, np.arange(5.0), np.array([1.0, 0, -1, -1, -1]), 5, C.byref(xi), C.byref(yi))


def test_numpy_port_vs_reference():
    """The second, independently written checker (oracle/rt_numpy.py) against the same goldens."""
    from oracle import rt_numpy
    g = load_golden("compare_cfg.npz")
    _check8(rt_numpy.shoot(0.0, D_PLANE, g["zf"], g["alpha"], 0.037, 0.0038), g["out8"], "numpy/compare")
    e = load_golden("edge_cfg.npz")
    for tag in ("q1nan", "tir", "off0", "offtx"):
        r_o, off, x_tx = e[tag + "_cfg"]
        _check8(rt_numpy.shoot(x_tx, D_PLANE, np.full(905, D_PLANE), e["alpha"], r_o, off), e[tag], "numpy/" + tag)
    s = load_golden("sweep_cfg.npz")
    for gi in range(0, 210, 13):
        o = rt_numpy.shoot(0.0, D_PLANE, np.full(905, D_PLANE), s["alpha"], *s["geoms"][gi])
        ref = s["target_x_tof"][gi]
        assert nan_equal_mask(o[6], ref[0]) and max_abs(o[6], ref[0]) < 1e-11


def test_lens_two_point_oracle_is_pinned_by_reference_rays():
    """Fermat <=> Snell: golden-section minimiser (oracle) on (A, F = pipe point of a reference ray) gives
    that ray's alpha and tof_1 + tof_2 (main_compare.py:514-515)."""
    g = load_golden("compare_cfg.npz")
    o, alpha, t4 = g["out8"], g["alpha"], g["tof4"]
    idx = np.nonzero(np.isfinite(o[2]))[0][::9]
    am = float(np.float64(50.62033040986099 * (np.pi / 180)))
    tt, al = cport.tt_lens([0.0], [D_PLANE], o[2][idx], o[3][idx], -am, am)
    assert np.max(np.abs(tt[0] - (t4[0] + t4[1])[idx])) < 1e-16
    assert np.max(np.abs(al[0] - alpha[idx])) < 1e-8


def test_oracle_root_find_reproduces_scan_hits():
    """orc_solve (bisection on x_land(alpha) = x_rx) vs the reference's database_2.csv: each tolerance hit of the
    grid scan is one of the element's roots, to the scan's own resolution (< 6e-8 s; offset-0 continuum excluded)."""
    rows = list(csv.reader(open(os.path.join(GOLDEN, "database_2.csv"))))[1:]
    db_hit = np.array([r[3] == "True" for r in rows]).reshape(210, 65)
    db_tof = np.array([float(r[4]) for r in rows]).reshape(210, 65)
    s = load_golden("sweep_cfg.npz")
    alpha, geoms, xe = s["alpha"], s["geoms"], s["x_elem"]
    checked = 0
    for gi in range(0, 210, 5):
        if not db_hit[gi].any() or abs(geoms[gi, 1]) < 1e-12:
            continue
        tt, ta, aa = cport.solve(0.0, D_PLANE, D_PLANE, alpha, xe, geoms[gi, 0], geoms[gi, 1])
        for e in np.nonzero(db_hit[gi])[0]:
            assert np.nanmin(np.abs(ta[e] - db_tof[gi, e])) < 6e-8
            checked += 1
        assert np.nanmin(tt) >= np.nanmin(ta) - 1e-18
    assert checked > 80


def test_planar_oracle_vs_50_digit_values():
    """The planar-layer solves cannot be pinned to the reference (it has no planar interfaces), but their
    arithmetic can: tests/golden/planar_mp.npz holds travel times from a 50-digit mpmath solve of Snell's law in
    the ray parameter p (tests/golden/make_planar_mp.py) — a formulation and a number system independent of the
    oracle's long-double bisection in q = tan(theta).  684 solves: cfg2 / cfg3 media, a 9-layer medium with
    near-critical rays, elements inside deeper layers.  Tolerance: 1 ulp of float64 (observed: bit-identical);
    the fp64 Newton CPU port used as bench.py's cpu_baseline: 1e-15 relative."""
    from oracle import cport
    g = load_golden("planar_mp.npz")
    for name in ("cfg2", "cfg3", "deep", "inner"):
        a = [g[f"{name}_{k}"] for k in ("z_if", "c", "xe", "ze", "xf", "zf")]
        ref = g[f"{name}_tt"]
        m = np.isfinite(ref)
        assert m.sum() >= 55
        for fn, tol in ((cport.tt_layers, 2.3e-16), (cport.tt_layers_newton, 1e-15)):
            tt = fn(*a)
            assert np.array_equal(np.isnan(tt), ~m), name
            assert np.max(np.abs(tt - ref)[m] / ref[m]) < tol, (name, fn.__name__)


def test_loop_structured_numpy_port_vs_reference():
    """oracle/rt_numpy_loop.py keeps the reference's per-ray Python loop (main_rt.py:384-393); same goldens, same tolerances."""
    from oracle import rt_numpy_loop
    g = load_golden("compare_cfg.npz")
    _check8(rt_numpy_loop.shoot(0.0, D_PLANE, g["zf"], g["alpha"], 0.037, 0.0038), g["out8"], "numpy-loop/compare")
    e = load_golden("edge_cfg.npz")
    for tag in ("q1nan", "tir", "off0", "offtx"):
        r_o, off, x_tx = e[tag + "_cfg"]
        _check8(rt_numpy_loop.shoot(x_tx, D_PLANE, np.full(905, D_PLANE), e["alpha"], r_o, off), e[tag], "numpy-loop/" + tag)


def test_oracle_vs_reference_on_inputs_its_drivers_never_use():
    """tests/golden/random_cfg.npz (make_golden_random.py ran the reference): non-uniform, narrow and DESCENDING launch-angle
    grids, a landing depth that differs from ray to ray, random geometries and transmit positions."""
    g = load_golden("random_cfg.npz")
    worst = 0.0
    for i in range(int(g["n_cases"])):
        tag = f"c{i:02d}"
        assert int(g[tag + "_raises"]) == 0
        n, r_outer, off, x_tx = g[tag + "_cfg"]
        o, _ = cport.shoot(float(x_tx), D_PLANE, g[tag + "_zf"], g[tag + "_alpha"], float(r_outer), float(off))
        _check8(o, g[tag + "_out8"], tag)
        worst = max(worst, max(max_abs(o[k], g[tag + "_out8"][k]) for k in range(8)))
    assert worst < POS_TOL

"""GPU parity: HIP forward trace + matcher (through the C ABI) vs the golden vectors captured
from the reference and vs the C oracle.  Tolerances: positions 1e-12 m, times 1e-15 s absolute
(the north-star bar is max|dt| < 1e-9 s; fp64 libm differences give ~1e-18 s in practice).
"""
import numpy as np
import pytest

from conftest import D_PLANE, load_golden, max_abs, nan_equal_mask

pytestmark = pytest.mark.gpu

POS_TOL = 1e-12     # metres (landing x amplifies ulp noise through 1/tan: observed ~1e-14)
TIME_TOL = 1e-15    # seconds (bar: 1e-9 s)


def _check8(out8, ref8, tag):
    for k in range(8):
        assert nan_equal_mask(out8[k], ref8[k]), f"{tag}: NaN mask differs in slot {k}"
        assert max_abs(out8[k], ref8[k]) < POS_TOL, f"{tag}: slot {k} off by {max_abs(out8[k], ref8[k])}"


def test_compare_config_matches_reference(rtus):
    g = load_golden("compare_cfg.npz")
    p = rtus.Params(r_outer=float(g["r_outer"]), pipe_offset=float(g["pipe_offset"]))
    res = rtus.shoot_rays(0.0, D_PLANE, g["zf"], g["alpha"], plot=False, params=p)
    out8 = np.stack([res[k] for k in rtus.KEYS])
    _check8(out8, g["out8"], "compare")
    b = rtus.shoot_batch([0.0], [D_PLANE], g["zf"], g["alpha"], params=p, want=("tof4", "tof", "land_x"))
    tof4 = b["tof4"][0, 0]
    assert nan_equal_mask(tof4, g["tof4"])
    assert max_abs(tof4, g["tof4"]) < TIME_TOL
    assert np.array_equal(b["land_x"][0, 0], out8[6], equal_nan=True)


def test_edge_cases_match_reference(rtus):
    e = load_golden("edge_cfg.npz")
    for tag in ("q1nan", "tir", "off0", "offtx"):
        r_o, off, x_tx = e[tag + "_cfg"]
        p = rtus.Params(r_outer=float(r_o), pipe_offset=float(off))
        res = rtus.shoot_rays(float(x_tx), D_PLANE, np.full(905, D_PLANE), e["alpha"], params=p)
        _check8(np.stack([res[k] for k in rtus.KEYS]), e[tag], tag)


def test_sweep_batched_matches_reference_and_database2(rtus):
    """All 210 geometries in ONE launch; landing x / tof per ray vs the reference capture, then
    the matcher vs database_2.csv (the reference's own known answers)."""
    s = load_golden("sweep_cfg.npz")
    alpha, geoms, x_elem = s["alpha"], s["geoms"], s["x_elem"]
    zf = np.full(alpha.size, D_PLANE)
    b = rtus.shoot_batch([0.0], [D_PLANE], zf, alpha, geoms, params=rtus.Params(),
                         want=("out8", "tof", "land_x"))
    ref = s["target_x_tof"]                       # [210, 2, 905]
    lx, tof = b["land_x"][:, 0], b["tof"][:, 0]
    assert nan_equal_mask(lx, ref[:, 0]) and nan_equal_mask(tof, ref[:, 1])
    assert max_abs(lx, ref[:, 0]) < 1e-11
    assert max_abs(tof, ref[:, 1]) < TIME_TOL
    for key in s.files:
        if key.startswith("full_"):
            r, o = key[6:].split("_o")
            gi = (int(r) - 1) * 21 + (int(o) + 10)
            _check8(b["out8"][gi, 0], s[key], key)
    hit, tof_hit, first = rtus.match_elements(lx, tof, x_elem, atol=1e-6)
    import csv, os
    from conftest import GOLDEN
    rows = list(csv.reader(open(os.path.join(GOLDEN, "database_2.csv"))))[1:]
    assert len(rows) == 210 * 65
    db_hit = np.array([r[3] == "True" for r in rows]).reshape(10, 21, 65)
    db_tof = np.array([float(r[4]) for r in rows]).reshape(10, 21, 65)
    got_hit = hit.reshape(10, 21, 65)
    got_tof = tof_hit.reshape(10, 21, 65)
    assert np.array_equal(got_hit, db_hit), f"{(got_hit != db_hit).sum()} hit flags differ from database_2.csv"
    assert np.max(np.abs(got_tof - db_tof)) < TIME_TOL


def test_all_tx_matches_reference(rtus):
    """Off-centre transmit: 65 tx in one launch vs 65 reference calls."""
    for tag in ("a", "b"):
        g = load_golden(f"alltx_{tag}.npz")
        p = rtus.Params(r_outer=float(g["r_outer"]), pipe_offset=float(g["pipe_offset"]))
        xe = g["x_elem"]
        b = rtus.shoot_batch(xe, np.full(xe.size, D_PLANE), np.full(905, D_PLANE), g["alpha"], params=p)
        for t in range(xe.size):
            _check8(b["out8"][0, t], g["out8"][t], f"alltx_{tag}[{t}]")


def test_nested_resolutions(rtus):
    g = load_golden("nest_cfg.npz")
    p = rtus.Params(r_outer=0.037, pipe_offset=0.0038)
    for n in (181, 1809, 3617):
        alpha = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)
        res = rtus.shoot_rays(0.0, D_PLANE, np.full(n, D_PLANE), alpha, params=p)
        _check8(np.stack([res[k] for k in rtus.KEYS]), g[f"n{n}"], f"n{n}")


def test_gpu_vs_oracle_random_geometries(rtus):
    """Seeded random geometries / tx positions / non-uniform alpha grids vs the C oracle."""
    from oracle import cport
    rng = np.random.default_rng(0)
    for trial in range(6):
        n = int(rng.integers(70, 1500))
        alpha = np.sort(rng.uniform(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n))
        zf = np.full(n, D_PLANE) + rng.uniform(-1e-3, 1e-3, n)
        geoms = np.stack([rng.uniform(0.01, 0.1, 3), rng.uniform(-0.01, 0.01, 3)], axis=1)
        xa = rng.uniform(-0.02, 0.02, 4)
        za = np.full(4, D_PLANE)
        b = rtus.shoot_batch(xa, za, zf, alpha, geoms, params=rtus.Params(), want=("out8", "tof4"))
        for gi in range(3):
            for t in range(4):
                o, _ = cport.shoot(xa[t], za[t], zf, alpha, geoms[gi, 0], geoms[gi, 1])
                _check8(b["out8"][gi, t], o, f"rand{trial}/{gi}/{t}")
                t4 = cport.tof4(xa[t], za[t], o)
                assert nan_equal_mask(b["tof4"][gi, t], t4)
                assert max_abs(b["tof4"][gi, t], t4) < TIME_TOL


def test_matcher_vs_oracle(rtus):
    from oracle import cport
    rng = np.random.default_rng(1)
    n, e = 4000, 257
    x_rx = np.sort(rng.uniform(-0.05, 0.05, e))
    land = rng.uniform(-0.06, 0.06, (3, n))
    land[:, ::7] = np.nan
    land[0, 5:40] = x_rx[10] + 5e-7            # many rays on one element: first index must win
    land[1, 100] = np.inf
    tof = rng.uniform(1e-5, 2e-4, (3, n))
    for atol in (1e-6, 1e-4, 2e-3):
        for xr in (x_rx, x_rx[::-1].copy(), rng.permutation(x_rx)):     # sorted + unsorted paths
            hit, th, first = rtus.match_elements(land, tof, xr, atol=atol)
            rh = rtus.ray_hits(land, xr, atol=atol)
            for row in range(3):
                t4 = np.zeros((4, n)); t4[0] = tof[row]
                oh, ot, of = cport.match(land[row], t4, xr, atol)
                assert np.array_equal(hit[row], oh)
                assert np.array_equal(first[row], of)
                assert np.array_equal(th[row], ot)
                assert np.array_equal(rh[row], cport.ray_hits(land[row], xr, atol))


def test_fast_math_mode_vs_reference(rtus):
    """RTUS_SHOOT_FAST_MATH (vector-form laws, no trigonometry) against the same reference captures.
    Stated tolerance for this mode: NaN masks identical on regular rays, |dx| < 1e-9 m, |dt| < 1e-12 s
    (bar 1e-9 s).  Exactly vertical rays (alpha = 0 from the centre element) are capped at the slope the
    reference's tan(pi/2) gives, so they agree too; only rays the reference decides by rounding NOISE (the
    tangent alpha = 0 ray of the q1nan fixture) may differ — that is what the default mode is for."""
    g = load_golden("compare_cfg.npz")
    p = rtus.Params(r_outer=float(g["r_outer"]), pipe_offset=float(g["pipe_offset"]))
    b = rtus.shoot_batch([0.0], [D_PLANE], g["zf"], g["alpha"], params=p, want=("out8", "tof4"), fast=True)
    o, t4 = b["out8"][0, 0], b["tof4"][0, 0]
    for k in range(8):
        assert nan_equal_mask(o[k], g["out8"][k])
        assert max_abs(o[k], g["out8"][k]) < 1e-9
    assert nan_equal_mask(t4, g["tof4"]) and max_abs(t4, g["tof4"]) < 1e-12
    s = load_golden("sweep_cfg.npz")
    zf = np.full(905, D_PLANE)
    b = rtus.shoot_batch([0.0], [D_PLANE], zf, s["alpha"], s["geoms"], params=rtus.Params(),
                         want=("tof", "land_x"), fast=True)
    ref = s["target_x_tof"]
    lx, tof = b["land_x"][:, 0], b["tof"][:, 0]
    mid = np.zeros_like(lx, dtype=bool)
    mid[np.abs(np.abs(s["geoms"][:, 1]) - s["geoms"][:, 0]) < 1e-12, 452] = True   # |offset| = r: alpha = 0 is tangent
    same = np.isnan(lx) == np.isnan(ref[:, 0])
    assert same[~mid].all()
    ok = ~np.isnan(ref[:, 0]) & ~np.isnan(lx) & ~mid
    assert np.max(np.abs(lx - ref[:, 0])[ok]) < 1e-9
    assert np.max(np.abs(tof - ref[:, 1])[ok]) < 1e-12
    hit, tof_hit, _ = rtus.match_elements(lx, tof, s["x_elem"], atol=1e-6)
    import csv, os
    from conftest import GOLDEN
    rows = list(csv.reader(open(os.path.join(GOLDEN, "database_2.csv"))))[1:]
    db_hit = np.array([r[3] == "True" for r in rows]).reshape(210, 65)
    db_tof = np.array([float(r[4]) for r in rows]).reshape(210, 65)
    assert (hit != db_hit).sum() <= 2                               # element hits sit on a 1e-6 m tolerance edge
    both = hit & db_hit
    assert np.max(np.abs(tof_hit - db_tof)[both]) < 1e-12
    ga = load_golden("alltx_a.npz")
    xe = ga["x_elem"]
    b = rtus.shoot_batch(xe, np.full(xe.size, D_PLANE), zf, ga["alpha"],
                         params=rtus.Params(r_outer=float(ga["r_outer"]), pipe_offset=float(ga["pipe_offset"])), fast=True)
    for k in range(8):
        assert nan_equal_mask(b["out8"][0, :, k], ga["out8"][:, k])
        assert max_abs(b["out8"][0, :, k], ga["out8"][:, k]) < 1e-9


def test_inputs_the_reference_drivers_never_use(rtus):
    """tests/golden/random_cfg.npz — the reference's shoot_rays run on non-uniform, narrow and descending launch-angle grids,
    per-ray landing depths, random geometries and transmit positions (make_golden_random.py): the drop-in takes the same
    arrays; reference-compatible arithmetic to 1e-12 m with identical NaN masks, vector-form arithmetic to 1e-9 m."""
    g = load_golden("random_cfg.npz")
    for i in range(int(g["n_cases"])):
        tag = f"c{i:02d}"
        n, r_outer, off, x_tx = g[tag + "_cfg"]
        p = rtus.Params(r_outer=float(r_outer), pipe_offset=float(off))
        res = rtus.shoot_rays(float(x_tx), D_PLANE, g[tag + "_zf"], g[tag + "_alpha"], plot=False, params=p)
        out8 = np.stack([res[k] for k in rtus.KEYS])
        _check8(out8, g[tag + "_out8"], tag)
        fast = rtus.shoot_batch([float(x_tx)], [D_PLANE], g[tag + "_zf"], g[tag + "_alpha"], params=p, fast=True)["out8"][0, 0]
        both = np.isfinite(fast[6]) & np.isfinite(g[tag + "_out8"][6])
        assert (np.isfinite(fast[6]) != np.isfinite(g[tag + "_out8"][6])).sum() <= 2, tag     # (a ray on a validity edge may flip)
        assert np.max(np.abs(fast[:, both] - g[tag + "_out8"][:, both])) < 1e-9, tag

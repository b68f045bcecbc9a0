"""GPU: the element loop's continuation (start extrapolated from the FOUR previous elements of a workgroup block; three in the lens kernel)
is only a guess — results must not depend on how the aperture is ordered.  Irregular apertures: duplicates,
reversed / shuffled order, clusters, depth changing inside a block, non-finite coordinates.  Checked against
the oracle and against one-element launches (no history at all).  Tolerances: 1e-13 s vs the oracle (its own
bisection tolerance), 1e-16 s between launch shapes (the Fermat expansion's truncation level).
"""
import numpy as np
import pytest

from conftest import D_PLANE

pytestmark = pytest.mark.gpu


def _apertures():
    rng = np.random.default_rng(11)
    reg = (np.arange(40) - 19.5) * 0.6e-3
    yield "reversed", reg[::-1].copy(), np.zeros(40)
    yield "shuffled", rng.permutation(reg), np.zeros(40)
    yield "duplicates", np.repeat(reg[::4], 4), np.zeros(40)
    yield "clusters", np.sort(np.concatenate([rng.normal(-0.01, 1e-5, 20), rng.normal(0.012, 2e-3, 20)])), np.zeros(40)
    yield "depth steps", reg, np.repeat([0.0, 0.001, 0.0, -0.002, 0.0015], 8)
    yield "depth alternates", reg, np.tile([0.0, 0.0005], 20)
    yield "pairs then jump", np.concatenate([reg[:2], reg[:2] + 0.03, reg[:36] * 3.0]), np.zeros(40)


@pytest.mark.parametrize("media", [([0.020], [2330.0, 1483.0]), ([0.010, 0.025], [1483.0, 5900.0, 2330.0])])
def test_planar_irregular_apertures(rtus, media):
    from oracle import cport
    z_if, c = media
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 37), np.linspace(0.004, 0.06, 23))
    xf, zf = xs.ravel(), zs.ravel()
    for name, xe, ze in _apertures():
        tt, it = rtus.travel_time_layers(z_if, c, xe, ze, xf, zf, return_iters=True)
        ref = cport.tt_layers(z_if, c, xe, ze, xf, zf)
        ok = zf[None, :] > ze[:, None]
        assert np.isnan(tt[~ok]).all(), name
        assert np.max(np.abs(tt - ref)[ok]) < 1e-13, name
        assert it.max() < 40, name
        for e in (0, 1, 2, 3, 17, 39):                       # the same element alone: no history
            one = rtus.travel_time_layers(z_if, c, xe[e:e + 1], ze[e:e + 1], xf, zf)
            assert np.nanmax(np.abs(one[0] - tt[e])) < 1e-16, (name, e)


def test_planar_non_finite_coordinates_stay_in_their_row_or_column(rtus):
    z_if, c = [0.02], [2330.0, 1483.0]
    xe = (np.arange(24) - 11.5) * 0.6e-3
    ze = np.zeros(24)
    xf, zf = np.linspace(-0.02, 0.02, 300), np.full(300, 0.045)
    good = rtus.travel_time_layers(z_if, c, xe, ze, xf, zf)
    bad_xe, bad_ze = xe.copy(), ze.copy()
    bad_xe[5], bad_xe[6], bad_ze[13] = np.nan, np.inf, np.nan
    bad_xf = xf.copy()
    bad_xf[100] = np.nan
    tt = rtus.travel_time_layers(z_if, c, bad_xe, bad_ze, bad_xf, zf)
    rows = np.ones(24, bool); rows[[5, 6, 13]] = False
    cols = np.ones(300, bool); cols[100] = False
    assert not np.isfinite(tt[~rows]).any() and np.isnan(tt[:, 100]).all()
    assert np.max(np.abs(tt[np.ix_(rows, cols)] - good[np.ix_(rows, cols)])) < 1e-16


def test_lens_irregular_apertures(rtus):
    rng = np.random.default_rng(5)
    reg = (np.arange(40) - 19.5) * 0.5e-3
    xs, zs = np.meshgrid(np.linspace(-0.004, 0.004, 31), np.linspace(0.03, 0.07, 17))
    xf, zf = xs.ravel(), zs.ravel()
    p = rtus.Params()
    base, abase = rtus.travel_time_lens(reg, np.full(40, D_PLANE), xf, zf, params=p, return_alpha=True)
    interior0 = np.abs(abase) < rtus.ALPHA_MAX - 1e-6
    for name, order in (("reversed", np.arange(40)[::-1]), ("shuffled", rng.permutation(40)),
                        ("duplicates", np.repeat(np.arange(0, 40, 4), 4))):
        tt, al = rtus.travel_time_lens(reg[order], np.full(40, D_PLANE), xf, zf, params=p, return_alpha=True)
        m = interior0[order]
        assert m.sum() > 2000
        assert np.max(np.abs(tt - base[order])[m]) < 1e-16, name
        assert np.max(np.abs(al - abase[order])[m]) < 1e-9, name
        t32 = rtus.travel_time_lens(reg[order], np.full(40, D_PLANE), xf, zf, params=p, dtype=np.float32)
        assert np.max(np.abs(t32.astype(np.float64) - base[order])[m]) < 2e-10, name
    # element depth changing inside a block (history restarts)
    ze = np.full(40, D_PLANE) + np.repeat([0.0, 1e-4, 0.0, -1e-4, 0.0], 8)
    tt = rtus.travel_time_lens(reg, ze, xf, zf, params=p)
    for e in (0, 8, 9, 16, 39):
        one = rtus.travel_time_lens(reg[e:e + 1], ze[e:e + 1], xf, zf, params=p)
        m = interior0[e]
        assert np.max(np.abs(one[0] - tt[e])[m]) < 1e-16, e


def test_lens_wide_alpha_interval_uses_generic_trig_and_agrees(rtus):
    """|alpha| <= 1 rad runs the polynomial sin/cos, a wider search interval the generic sincos: same minima."""
    p = rtus.Params()
    xe = (np.arange(12) - 5.5) * 0.5e-3
    xs, zs = np.meshgrid(np.linspace(-0.003, 0.003, 21), np.linspace(0.035, 0.065, 11))
    xf, zf = xs.ravel(), zs.ravel()
    ze = np.full(12, D_PLANE)
    t0, a0 = rtus.travel_time_lens(xe, ze, xf, zf, params=p, return_alpha=True)
    t1, a1 = rtus.travel_time_lens(xe, ze, xf, zf, params=p, return_alpha=True, alpha_lo=-1.02, alpha_hi=1.02)
    m = (np.abs(a0) < 0.85) & np.isfinite(t1)
    assert m.sum() > 1500
    assert np.max(np.abs(t0 - t1)[m]) < 1e-16
    assert np.max(np.abs(a0 - a1)[m]) < 1e-9
    f0 = rtus.travel_time_lens(xe, ze, xf, zf, params=p, dtype=np.float32)
    f1 = rtus.travel_time_lens(xe, ze, xf, zf, params=p, dtype=np.float32, alpha_lo=-1.02, alpha_hi=1.02)
    assert np.max(np.abs(f0.astype(np.float64) - f1)[m]) < 2e-10

"""GPU: planar-layer Fermat travel times (BASELINE configs 2/3/5) vs the long-double bisection
oracle and closed forms.  NOT pinned by the reference (it has no planar interfaces): "parity
unpinned".  Tolerance: |dt| < 1e-13 s absolute (times are ~1e-5 s; bar is 1e-9 s).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-13


def _grid(nx, nz, x0, x1, z0, z1):
    xs, zs = np.linspace(x0, x1, nx), np.linspace(z0, z1, nz)
    X, Z = np.meshgrid(xs, zs)
    return X.ravel(), Z.ravel()


def test_config2_one_interface_vs_oracle(rtus):
    from oracle import cport
    xe = (np.arange(128) - 63.5) * 0.6e-3
    ze = np.zeros(128)
    xf, zf = _grid(128, 128, -0.02, 0.02, 0.025, 0.065)
    for c in ((2330.0, 1483.0), (1483.0, 5900.0)):
        tt, it = rtus.travel_time_layers([0.020], c, xe, ze, xf, zf, return_iters=True)
        ref = cport.tt_layers([0.020], c, xe[::9], ze[::9], xf, zf)
        assert np.max(np.abs(tt[::9] - ref)) < TOL
        assert it.max() < 40


def test_config3_two_interfaces_and_targets_in_every_layer(rtus):
    from oracle import cport
    xe = (np.arange(64) - 31.5) * 0.3e-3
    ze = np.zeros(64)
    xf, zf = _grid(96, 80, -0.03, 0.03, 0.001, 0.060)      # targets above, between and below interfaces
    z_if, c = [0.010, 0.025], [2330.0, 1483.0, 5900.0]
    tt = rtus.travel_time_layers(z_if, c, xe, ze, xf, zf)
    ref = cport.tt_layers(z_if, c, xe, ze, xf, zf)
    assert np.max(np.abs(tt - ref)) < TOL


def test_closed_forms(rtus):
    xe, ze = np.array([-0.01, 0.0, 0.004]), np.array([0.0, 0.001, 0.0])
    xf, zf = _grid(33, 17, -0.03, 0.03, 0.012, 0.05)
    # equal speeds -> straight line
    tt = rtus.travel_time_layers([0.01, 0.02], [1500.0, 1500.0, 1500.0], xe, ze, xf, zf)
    ref = np.hypot(xf[None, :] - xe[:, None], zf[None, :] - ze[:, None]) / 1500.0
    assert np.max(np.abs(tt - ref)) < 1e-15
    # no interface at all
    tt0 = rtus.travel_time_layers([], [1500.0], xe, ze, xf, zf)
    assert np.max(np.abs(tt0 - ref)) < 1e-15
    # vertical incidence: sum h_i / c_i
    t = rtus.travel_time_layers([0.01, 0.03], [1000.0, 2000.0, 4000.0], [0.002], [0.0], [0.002], [0.05])
    assert abs(t[0, 0] - (0.01 / 1000 + 0.02 / 2000 + 0.02 / 4000)) < 1e-18
    # target not below the source -> NaN
    t = rtus.travel_time_layers([0.01], [1000.0, 2000.0], [0.0], [0.005], [0.001], [0.005])
    assert np.isnan(t[0, 0])


def test_fermat_stationarity_and_snell(rtus):
    """The returned time is the minimum over interface crossing points (1 interface)."""
    c0, c1, zi = 2330.0, 1483.0, 0.02
    xe, ze, xf, zf = 0.0, 0.0, 0.013, 0.047
    t = rtus.travel_time_layers([zi], [c0, c1], [xe], [ze], [xf], [zf])[0, 0]
    xs = np.linspace(xe, xf, 2_000_001)
    T = np.hypot(xs - xe, zi - ze) / c0 + np.hypot(xf - xs, zf - zi) / c1
    assert t <= T.min() + 1e-18
    assert T.min() - t < 1e-15


def test_many_layers_and_grazing(rtus):
    from oracle import cport
    rng = np.random.default_rng(3)
    z_if = np.cumsum(rng.uniform(0.002, 0.01, 8))
    c = rng.uniform(1000.0, 6500.0, 9)
    xe, ze = rng.uniform(-0.05, 0.05, 16), rng.uniform(-0.004, 0.0015, 16)
    xf = rng.uniform(-0.4, 0.4, 500)                       # large offsets: near-critical rays
    zf = rng.uniform(0.002, z_if[-1] + 0.02, 500)
    tt, it = rtus.travel_time_layers(z_if, c, xe, ze, xf, zf, return_iters=True)
    ref = cport.tt_layers(z_if, c, xe, ze, xf, zf)
    assert np.max(np.abs(tt - ref) / ref) < 1e-12
    assert it.max() < 60


def test_sources_inside_deeper_layers_and_same_layer_targets(rtus):
    """Elements below the first interface(s), targets in the element's own layer, extreme offsets."""
    from oracle import cport
    z_if, c = [0.005, 0.012, 0.020], [1500.0, 3200.0, 1480.0, 5900.0]
    xe = np.array([-0.01, 0.0, 0.003, 0.02])
    ze = np.array([0.0, 0.006, 0.0125, 0.0199])               # layer 0, 1, 2, 2 (just above an interface)
    rng = np.random.default_rng(21)
    xf = np.concatenate([rng.uniform(-0.05, 0.05, 300), [0.0, 0.003, 5.0, -5.0]])
    zf = np.concatenate([rng.uniform(0.0005, 0.03, 300), [0.0061, 0.0126, 0.025, 0.0201]])
    tt, it = rtus.travel_time_layers(z_if, c, xe, ze, xf, zf, return_iters=True)
    ref = cport.tt_layers(z_if, c, xe, ze, xf, zf)
    ok = zf[None, :] > ze[:, None]
    assert np.isnan(tt[~ok]).all()                            # targets not below the source
    assert np.max(np.abs(tt - ref)[ok] / ref[ok]) < 1e-12
    assert it.max() < 60


def test_gpu_vs_50_digit_values(rtus):
    """GPU solver against the 50-digit mpmath values (tests/golden/planar_mp.npz, see make_planar_mp.py): the check
    that does not go through the C oracle at all.  Tolerance |dt| <= 1e-16 s + 5e-12 t: the solver stops Newton
    within 3e-4 of the root and removes the rest to second order, so its remainder is O((3e-4)^3) relative —
    3e-17 s on the ~3e-5 s times of configs 2/3 (the bar is 1e-9 s)."""
    from conftest import load_golden
    g = load_golden("planar_mp.npz")
    for name in ("cfg2", "cfg3", "deep", "inner"):
        a = [g[f"{name}_{k}"] for k in ("z_if", "c", "xe", "ze", "xf", "zf")]
        ref = g[f"{name}_tt"]
        m = np.isfinite(ref)
        tt = rtus.travel_time_layers(*a)
        assert np.array_equal(np.isnan(tt), ~m), name
        err = np.abs(tt - ref)[m]
        assert np.all(err <= 1e-16 + 5e-12 * ref[m]), (name, float(err.max()), float((err / ref[m]).max()))

"""GPU: two-point Fermat solves through the reference's curved lens surface (BASELINE config 4) and the
FMC reflector table (config 5).

Pin to the reference: Fermat <=> Snell, so for every forward-traced reference ray the solve
(A = element, F = that ray's pipe hit point) must return the ray's launch angle alpha and
tof_1 + tof_2 (main_compare.py:514-515) — checked against the goldens captured from the reference.
Tolerances: fp64 |dt| < 1e-15 s, |dalpha| < 1e-10 rad; fp32 |dt| < 2e-10 s (measured ~3e-11; the bar
for fp32 is stated separately, SURVEY 7 'fp32 (config 4)').
"""
import numpy as np
import pytest

from conftest import D_PLANE, load_golden

pytestmark = pytest.mark.gpu


def _ok_rays(g):
    o = g["out8"]
    return np.nonzero(np.isfinite(o[2]) & np.isfinite(o[3]))[0]


def test_lens_solver_reproduces_reference_rays(rtus):
    g = load_golden("compare_cfg.npz")
    o, alpha, t4 = g["out8"], g["alpha"], g["tof4"]
    idx = _ok_rays(g)
    tt, al = rtus.travel_time_lens([0.0], [D_PLANE], o[2][idx], o[3][idx], params=rtus.Params(), return_alpha=True)
    assert np.max(np.abs(tt[0] - (t4[0] + t4[1])[idx])) < 1e-15
    assert np.max(np.abs(al[0] - alpha[idx])) < 1e-10


def test_lens_solver_off_centre_elements_reference_rays(rtus):
    g = load_golden("alltx_a.npz")
    xe = g["x_elem"]
    for t in (0, 17, 40, 64):
        o = g["out8"][t]
        idx = np.nonzero(np.isfinite(o[2]))[0]
        tt, al = rtus.travel_time_lens([xe[t]], [D_PLANE], o[2][idx], o[3][idx], params=rtus.Params(), return_alpha=True)
        t12 = (np.hypot(xe[t] - o[0], D_PLANE - o[1]) / 6400.0 + np.hypot(o[0] - o[2], o[1] - o[3]) / 1483.0)[idx]
        assert np.max(np.abs(tt[0] - t12)) < 1e-15
        assert np.max(np.abs(al[0] - g["alpha"][idx])) < 1e-10


def test_lens_points_along_reference_rays_fp64_fp32_vs_oracle(rtus):
    """Targets ON refracted reference rays (the region the lens actually insonifies: elsewhere the
    least-time point is the aperture edge) at several depths, 9 elements; GPU fp64 / fp32 vs the
    golden-section oracle."""
    from oracle import cport
    g = load_golden("alltx_a.npz")
    xe_all = g["x_elem"]
    sel = [0, 8, 16, 24, 32, 40, 48, 56, 64]
    rng = np.random.default_rng(7)
    for t in sel:
        o = g["out8"][t]
        ok = np.nonzero(np.isfinite(o[2]))[0]
        rays = rng.choice(ok, 60, replace=False)
        s_frac = rng.uniform(0.15, 0.95, rays.size)
        xf = o[0][rays] + s_frac * (o[2][rays] - o[0][rays])
        zf = o[1][rays] + s_frac * (o[3][rays] - o[1][rays])
        ref, aref = cport.tt_lens([xe_all[t]], [D_PLANE], xf, zf, -rtus.ALPHA_MAX, rtus.ALPHA_MAX)
        tt, al = rtus.travel_time_lens([xe_all[t]], [D_PLANE], xf, zf, params=rtus.Params(), return_alpha=True)
        assert np.max(np.abs(tt - ref)) < 1e-15
        assert np.max(np.abs(al[0] - g["alpha"][rays])) < 1e-9     # the ray's own launch angle
        t32 = rtus.travel_time_lens([xe_all[t]], [D_PLANE], xf, zf, params=rtus.Params(), dtype=np.float32)
        assert t32.dtype == np.float32
        assert np.max(np.abs(t32.astype(np.float64) - ref)) < 2e-10


def test_fmc_reflector_table_vs_two_leg_minimisation(rtus):
    """Config 5: unfolded-stack table vs min over reflection points of (down leg + up leg), both legs by the oracle."""
    from oracle import cport
    z_if, c, z_r = [0.008, 0.02], [2330.0, 1483.0, 5900.0], 0.035
    x = (np.arange(24) - 11.5) * 1.5e-3
    tt = rtus.fmc_table_layers(z_if, c, x, x, z_r)
    assert tt.shape == (24, 24) and np.allclose(tt, tt.T, atol=1e-18)
    xr = np.linspace(-0.03, 0.03, 24001)
    legs = cport.tt_layers_newton(z_if, c, x, np.zeros(24), xr, np.full(xr.size, z_r))   # [24, n_xr]
    for i in (0, 5, 23):
        for j in (0, 11, 23):
            brute = np.min(legs[i] + legs[j])
            assert tt[i, j] <= brute + 1e-16 and brute - tt[i, j] < 1e-13    # solver truncation error ~3e-17 s

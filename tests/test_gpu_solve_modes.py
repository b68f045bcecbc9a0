"""GPU: the two bracket-finding paths of rtus_solve and the order independence of its answers.

Brackets come from the grid trace as pair masks (apertures up to 512 elements) or from a scan of the landing points inside
the solve kernel (larger apertures).  A bracket is refined by one lane whose arithmetic does not depend on which other
brackets share its wave, so the SAME element must get the SAME BITS whatever the path, the order of the aperture or the
company it is solved in.  (Checker for the values themselves: tests/test_gpu_solve.py against the oracle's bisection.)"""
import numpy as np
import pytest

from conftest import D_PLANE, load_golden

pytestmark = pytest.mark.gpu


def _solve(rtus, x_rx, geoms, txs, alpha):
    return rtus.solve_travel_times(txs, np.full(len(txs), D_PLANE), x_rx, alpha, geoms, params=rtus.Params(), all_roots=True)


def test_shuffled_aperture_gives_the_same_bits(rtus):
    s = load_golden("sweep_cfg.npz")
    alpha, xe = s["alpha"], s["x_elem"]
    geoms = s["geoms"][[5, 64, 133, 201]]
    txs = np.array([0.0, 0.0071])
    ref = _solve(rtus, xe, geoms, txs, alpha)
    perm = np.random.default_rng(7).permutation(xe.size)
    got = _solve(rtus, xe[perm], geoms, txs, alpha)
    for a, b in zip(ref, got):
        assert np.array_equal(a[:, :, perm], b, equal_nan=True)
    assert (ref[4] > 0).sum() > 100


def test_scan_path_and_mask_path_agree_bit_for_bit(rtus):
    s = load_golden("sweep_cfg.npz")
    alpha = s["alpha"]
    geoms = s["geoms"][[17, 88, 160]]
    txs = np.array([-0.004, 0.0])
    x_big = np.linspace(-0.03, 0.03, 700)                      # > 512 elements: lock-step scan inside the solve kernel
    big = _solve(rtus, x_big, geoms, txs, alpha)
    for lo in (0, 350):                                        # the same elements, 350 at a time: pair masks from the grid trace
        part = _solve(rtus, x_big[lo:lo + 350], geoms, txs, alpha)
        for a, b in zip(big, part):
            assert np.array_equal(a[:, :, lo:lo + 350], b, equal_nan=True)
    assert (big[4] > 0).sum() > 1000


def test_duplicate_and_single_elements(rtus):
    s = load_golden("sweep_cfg.npz")
    alpha, geoms = s["alpha"], s["geoms"][[40]]
    x = np.array([0.0031, 0.0031, -0.002, 0.0031])
    tt, ar, ta, aa, nr = _solve(rtus, x, geoms, np.array([0.0]), alpha)
    assert np.array_equal(ta[0, 0, 0], ta[0, 0, 1], equal_nan=True) and np.array_equal(ta[0, 0, 0], ta[0, 0, 3], equal_nan=True)
    one = _solve(rtus, x[2:3], geoms, np.array([0.0]), alpha)
    assert np.array_equal(one[2][0, 0, 0], ta[0, 0, 2], equal_nan=True)


def test_long_grid_with_block_boundary_pairs(rtus):
    """N = 4097 rays: 65 blocks of 64 pairs; brackets that straddle two blocks come from the solve kernel's own check."""
    from oracle import cport
    n = 4097
    alpha = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)
    x = np.linspace(-0.019, 0.019, 130)
    tt, ar, ta, aa, nr = rtus.solve_travel_times([0.0021], [D_PLANE], x, alpha, [[0.037, 0.0038]], params=rtus.Params(), all_roots=True)
    otm, ota, oaa = cport.solve(0.0021, D_PLANE, D_PLANE, alpha, x, 0.037, 0.0038)
    same = np.isfinite(ota).sum(1) == nr[0, 0]
    assert same.mean() > 0.97
    m = same[:, None] & np.isfinite(ota)
    assert np.max(np.abs(ta[0, 0] - ota)[m]) < 1e-13
    # some bracket's lower ray is the last ray of a 64-ray block
    idx = np.searchsorted(alpha, aa[0, 0][np.isfinite(aa[0, 0])]) - 1
    assert np.any(idx % 64 == 63)


def test_one_launch_three_launches_and_one_lane(rtus):
    """Round 4: a call of the reference's size runs as ONE kernel (a workgroup per row, landing points and pair masks in LDS) with
    three lanes per bracket; RTUS_SOLVE_THREE_LAUNCHES keeps the grid trace and the refinement apart — the refinement is the same
    function, so the SAME BITS; RTUS_SOLVE_ONE_LANE is the other iteration: the same roots to the solver's tolerances."""
    s = load_golden("sweep_cfg.npz")
    alpha, xe = s["alpha"], s["x_elem"]
    geoms = s["geoms"][::7]
    txs = np.array([0.0, -0.0093])
    za = np.full(2, D_PLANE)
    one = rtus.solve_travel_times(txs, za, xe, alpha, geoms, params=rtus.Params(), all_roots=True)
    three = rtus.solve_travel_times(txs, za, xe, alpha, geoms, params=rtus.Params(), all_roots=True, three_launches=True)
    for a, b in zip(one, three):
        assert np.array_equal(a, b, equal_nan=True)
    lane1 = rtus.solve_travel_times(txs, za, xe, alpha, geoms, params=rtus.Params(), all_roots=True, one_lane=True)
    same = one[4] == lane1[4]                                  # root counts (a branch that ends inside a bracket may differ)
    assert same.mean() > 0.995 and (one[4] > 0).sum() > 1000
    m = same[..., None] & np.isfinite(one[2]) & np.isfinite(lane1[2])
    assert np.max(np.abs(one[2] - lane1[2])[m]) < 1e-13
    # alpha: the transmitting element sees x_land nearly flat (|dx/dalpha| ~ 1e-4 m/rad): compare the landing error it implies instead
    assert np.nanmax(np.abs(one[0] - lane1[0])[same]) < 1e-13
    fast1 = rtus.solve_travel_times(txs, za, xe, alpha, geoms, params=rtus.Params(), all_roots=True, fast=True)
    fast3 = rtus.solve_travel_times(txs, za, xe, alpha, geoms, params=rtus.Params(), all_roots=True, fast=True, three_launches=True)
    # (the vector-form trace has wave-level fast paths: its last bits may depend on a lane's wave-mates — tolerance, not bits)
    assert (fast1[4] == fast3[4]).mean() > 0.999
    mf = (fast1[4] == fast3[4])[..., None] & np.isfinite(fast1[2]) & np.isfinite(fast3[2])
    assert np.max(np.abs(fast1[2] - fast3[2])[mf]) < 1e-13


@pytest.mark.parametrize("n,n_rx", [(70, 1), (1000, 64), (1024, 128), (905, 65), (129, 127)])
def test_one_launch_path_ragged_sizes(rtus, n, n_rx):
    """The one-launch kernel at the edges of its domain (grid of up to 1,024 rays = one workgroup, up to 128 elements, the last 64-ray
    block partly filled, a single element, an unsorted aperture with a duplicate and a NaN): the same bits as the separate launches,
    the same roots as the oracle's bisection."""
    from oracle import cport
    rng = np.random.default_rng(n * 131 + n_rx)
    alpha = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)
    x = rng.uniform(-0.019, 0.019, n_rx)
    if n_rx > 4:
        x[3] = x[1]
        x[2] = np.nan
    geoms = np.array([[0.037, 0.0038], [0.02, -0.006], [0.08, 0.001]])
    txs = np.array([0.0021, -0.011])
    za = np.full(2, D_PLANE)
    one = rtus.solve_travel_times(txs, za, x, alpha, geoms, params=rtus.Params(), all_roots=True)
    three = rtus.solve_travel_times(txs, za, x, alpha, geoms, params=rtus.Params(), all_roots=True, three_launches=True)
    for a, b in zip(one, three):
        assert np.array_equal(a, b, equal_nan=True)
    if n_rx > 4:
        assert one[4][:, :, 2].max() == 0 and np.isnan(one[0][:, :, 2]).all()          # the NaN element has no ray path
        assert np.array_equal(one[2][:, :, 1], one[2][:, :, 3], equal_nan=True)         # duplicates: the same roots
    ok = np.isfinite(x)
    for g in range(3):
        for t in range(2):
            otm, ota, oaa = cport.solve(txs[t], D_PLANE, D_PLANE, alpha, x[ok], geoms[g, 0], geoms[g, 1])
            same = np.isfinite(ota).sum(1) == one[4][g, t][ok]
            assert same.mean() > 0.9 or n < 100
            m = same[:, None] & np.isfinite(ota)
            assert np.max(np.abs(one[2][g, t][ok] - ota)[m], initial=0) < 1e-13

"""Fuzz of the root-finding solve (rtus_solve: the one-launch row kernel with three lanes per bracket for small calls, the separate
launches, the one-lane iteration) against the oracle's bisection (orc_solve: the same bracketing rule, plain bisection): random pipe
geometries incl. centred and tangent pipes, transmit points, apertures (1 .. 128 elements, sorted or not, with duplicates), uniform
launch-angle grids of 64 .. 1,024 rays.  Per (geometry, tx, element): the root counts must agree except where a branch of x_land ends
inside a bracket (counted, <= 1 % of the elements), and where they agree every root must be within 1e-13 s and within
1e-11 rad (one-lane iteration: 5e-11, see below) + 5e-12 m / |dx_land/dalpha| (the GPU's and the oracle's traces differ by up to a few 1e-12 m in the landing point of an
ill-conditioned ray — their trigonometry differs in the last bit — and where x_land is flat that is all alpha can be known to; roots on stretches flatter than 1e-3 m/rad are compared in time only).

    gpurun -- python scripts/fuzz_solve.py [n_trials] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtus  # noqa: E402
from oracle import cport  # noqa: E402

D = float(np.float64(0.12156646438729327) + np.float64(0.08843353561270673))
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
worst_t = worst_a = worst_t1 = worst_a1 = 0.0
n_el = n_count_diff = n_roots = n_flat = 0
by_kind = {}
t0 = time.time()
for trial in range(trials):
    n = int(rng.choice([rng.integers(64, 200), rng.integers(200, 700), rng.integers(700, 1025), 905]))
    n_rx = int(rng.choice([1, rng.integers(2, 20), rng.integers(20, 129), 65]))
    alpha = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)
    x = rng.uniform(-0.02, 0.02, n_rx)
    if rng.random() < 0.5:
        x = np.sort(x)
    if n_rx > 3 and rng.random() < 0.3:
        x[2] = x[0]
    G, T = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    r = rng.uniform(0.008, 0.11, G)
    off = rng.uniform(-0.015, 0.015, G)
    kind = rng.integers(0, 4)
    if kind == 0: off[0] = 0.0
    if kind == 1: off[0] = r[0] * rng.choice([-1.0, 1.0]) * 0.999
    geoms = np.stack([r, off], axis=1)
    txs = np.concatenate([[0.0], rng.uniform(-0.019, 0.019, T - 1)]) if rng.random() < 0.5 else rng.uniform(-0.019, 0.019, T)
    za = np.full(T, D)
    mode = ("default", "three_launches", "one_lane")[trial % 3]
    fast = trial % 7 == 6                                      # the vector-form trace: another arithmetic than the oracle's — 1e-11 s, 1e-9 rad
    # (alpha: 1e-11 rad for the three-lane scheme; the one-lane iteration applies an untaken interpolation step of up to 1e-8 rad whose
    #  slope comes from points up to a grid step apart, and is held to 5e-11.  Until this script measured it, that iteration's straggler
    #  rescue accepted a zero from a triple as wide as the step that led to it: 1.6e-9 rad / 1.8e-12 s where the triple straddled a kink
    #  of x_land on a coarse polyline; its triples are now at most 2e-8 rad wide, as the three-lane scheme's)
    tol_t, tol_a, tol_cnt = (1e-11, 1e-9, 0.02) if fast else ((1e-13, 5e-11, 0.0) if mode == "one_lane" else (1e-13, 1e-11, 0.0))
    kw = {"three_launches": mode == "three_launches", "one_lane": mode == "one_lane", "fast": fast}
    tt, ar, ta, aa, nr = rtus.solve_travel_times(txs, za, x, alpha, geoms, params=rtus.Params(), all_roots=True, **kw)
    for g in range(G):
        for t in range(T):
            otm, ota, oaa = cport.solve(txs[t], D, D, alpha, x, geoms[g, 0], geoms[g, 1])
            same = np.isfinite(ota).sum(1) == nr[g, t]
            n_el += n_rx
            n_count_diff += 0 if fast else int((~same).sum())
            if fast and (~same).mean() > tol_cnt + 2.0 / n_rx:
                print(f"ROOT COUNTS (vector form) differ on {int((~same).sum())} of {n_rx}: trial {trial} mode {mode} geom {geoms[g]} tx {txs[t]}"); sys.exit(1)
            key = (int(kind) if g == 0 else 9, mode, bool(txs[t] == 0.0))
            by_kind[key] = by_kind.get(key, 0) + int((~same).sum())
            if (~same).sum() > 0.3 * n_rx and os.environ.get("FUZZ_VERBOSE"):
                e = int(np.flatnonzero(~same)[0])
                print(f"  counts differ on {int((~same).sum())}/{n_rx}: trial {trial} mode {mode} n={n} geom {geoms[g].tolist()} tx {txs[t]!r}; element {x[e]!r}: gpu {int(nr[g, t][e])} oracle {int(np.isfinite(ota[e]).sum())}")
            m = same[:, None] & np.isfinite(ota)
            if not m.any():
                continue
            if not np.array_equal(np.isnan(ta[g, t][same]), np.isnan(ota[same])):
                print(f"ROOT SLOTS DIFFER trial {trial} mode {mode} geom {geoms[g]} tx {txs[t]}"); sys.exit(1)
            dt = np.abs(ta[g, t] - ota)[m]
            da = np.abs(aa[g, t] - oaa)[m]
            # conditioning of alpha: slope of x_land between the grid rays around the root (from the oracle's own trace)
            o8 = cport.shoot(txs[t], D, np.full(n, D), alpha, geoms[g, 0], geoms[g, 1])[0][6]
            j = np.clip(np.searchsorted(alpha, oaa[m]) - 1, 0, n - 2)
            slope = np.abs(o8[j + 1] - o8[j]) / (alpha[1] - alpha[0])
            flat = ~(slope > 1e-3)
            n_roots += int(m.sum()); n_flat += int(flat.sum())
            allow = tol_a + 5e-12 / np.maximum(slope, 1e-3)
            if not fast and mode != "one_lane":
                worst_t = max(worst_t, float(dt.max()))
                if (~flat).any():
                    worst_a = max(worst_a, float((da / allow)[~flat].max()))
            if not fast and mode == "one_lane":
                worst_t1 = max(worst_t1, float(dt.max()))
                if (~flat).any():
                    worst_a1 = max(worst_a1, float((da / allow)[~flat].max()))
            if dt.max() > tol_t or ((~flat).any() and (da / allow)[~flat].max() > 1.0):
                k = int(np.argmax(np.where(flat, 0, da / allow))) if ((~flat).any() and (da / allow)[~flat].max() > 1.0) else int(np.argmax(dt))
                print(f"MISMATCH trial {trial} mode {mode} fast={fast} n={n} n_rx={n_rx} geom {geoms[g].tolist()} tx {txs[t]!r}: |dt| {dt.max():.2e} s, "
                      f"|dalpha| {da[~flat].max() if (~flat).any() else 0:.2e} rad (root alpha {oaa[m][k]!r}, slope {slope[k]:.2e})")
                sys.exit(1)
    if trial % 20 == 19:
        print(f"trial {trial + 1}/{trials}: {n_el} elements, {n_roots} roots compared ({n_flat} on flat stretches of x_land), root counts differ "
              f"on {n_count_diff}, worst |dt| {worst_t:.2e} s, |dalpha| / allowed {worst_a:.3f}, {time.time() - t0:.0f} s", flush=True)
print('count differences by (kind of geometry 0 [0 centred, 1 tangent, 2-3 random; 9 = other geometries], mode, tx == 0):', by_kind)
if n_count_diff > 0.01 * n_el:
    print(f"ROOT COUNTS differ on {n_count_diff} of {n_el} elements"); sys.exit(1)
print(f"OK: {trials} trials (default / three_launches / one_lane in turn), {n_el} elements, {n_roots} roots compared ({n_flat} on flat stretches), "
      f"root counts differ on {n_count_diff} elements, worst |dt| {worst_t:.2e} s, worst |dalpha| / (1e-11 rad + 5e-12 m / slope) {worst_a:.3f} "
      f"(three lanes per bracket); one lane per bracket: worst |dt| {worst_t1:.2e} s, worst |dalpha| / (5e-11 rad + 5e-12 m / slope) {worst_a1:.3f}")

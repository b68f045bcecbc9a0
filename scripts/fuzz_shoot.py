"""Fuzz of the forward trace (reference-compatible mode) against the C oracle: random ray counts (2 .. 9000: all
box-hierarchy depths, ragged leaves), pipe radii / offsets including offset = 0 (every ray retraces itself) and
|offset| = r (tangent pipes), off-centre elements, uniform and non-uniform launch-angle grids, per-ray landing depths.

Criterion (hard, per ray and per output): |gpu - oracle| <= 1e-12 m + 16 x spread, where `spread` is the oracle's OWN
sensitivity of that output: the largest change it shows over fourteen re-runs — ten with every trigonometric result
moved by 0 / +-1 / +2 units in the last place in fixed random patterns (rt_oracle.c, orc_set_trig_noise: what calling a
different math library does, which is exactly how the GPU differs from the CPU), four with the launch-angle grid, the
element position and the pipe offset nudged by one ulp.  A well-conditioned ray has spread ~1e-16 m and
must agree to 1e-12 m; a ray the reference's arithmetic itself decides by rounding (a near-vertical line through the
quadratic pipe formula, a chord crossed at the critical angle) is allowed what the reference allows itself, and nothing
more — a wrong polyline segment, branch or root is off by >= 1e-8 m on rays whose spread is 1e-17.  NaN masks must be
identical on every ray whose NaN status the oracle keeps under all nudges (a ray whose own NaN status flips with one ulp
of its input is rounding-decided in the reference itself and is only counted); in a trace where U rays flip, U/4 further
disagreements are tolerated (the degenerate continuum: offset 0, element on the axis, random grid with near-duplicate
vertices — uniform grids, the reference's own, have U = 0 there); a trace that exceeds this is looked at again with 48 more
noise patterns (the fourteen variants sample a degenerate trace's rounding-decided rays thinly) under the same rule.

    gpurun -- python scripts/fuzz_shoot.py [n_trials] [seed]
    python scripts/fuzz_shoot.py --selftest [...]   the same run, but every 97th finite ray of the GPU result is moved
        onto the NEXT polyline chord (an off-by-one segment index): the run must FAIL (exit code 3 = caught).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtus  # noqa: E402
from oracle import cport  # noqa: E402

D = float(np.float64(0.12156646438729327) + np.float64(0.08843353561270673))
ABS_TOL, K_SPREAD = 1e-12, 16.0


def nudge(v, k):
    """v moved by k units in the last place (k may be negative); exact zeros move by k * 2^-56"""
    v = np.asarray(v, dtype=np.float64)
    out = v.copy()
    for _ in range(abs(k)):
        out = np.nextafter(out, np.inf if k > 0 else -np.inf)
    return np.where(v == 0.0, k * 2.0 ** -56, out)            # ~1 ulp of the lengths involved (0.1 m)


NOISE_MODES = [int(v) for v in np.random.default_rng(20260).integers(1, 2 ** 32, 10)]
MORE_NOISE_MODES = [int(v) for v in np.random.default_rng(20261).integers(1, 2 ** 32, 48)]   # second look at a degenerate trace


def nan_stable_under(modes, xa, za, zf, alpha, geoms):
    """[G, T, n]: the ray keeps the NaN pattern of its eight outputs under each of these trigonometric noise patterns"""
    base = np.isnan(cport.shoot_batch(xa, za, zf, alpha, geoms))
    stable = np.ones(base.shape, dtype=bool)
    try:
        for mode in modes:
            cport.set_trig_noise(mode)
            stable &= np.isnan(cport.shoot_batch(xa, za, zf, alpha, geoms)) == base
    finally:
        cport.set_trig_noise(0)
    return stable.all(axis=2)


def oracle_with_spread(xa, za, zf, alpha, geoms):
    """-> base [G, T, 8, n], spread [G, T, 8, n] (max |variant - base|), stable [G, T, 8, n] (NaN status kept by every variant).
    Variants: every trigonometric result of the oracle moved by 0 / +-1 / +2 ulp in ten fixed random patterns (what another
    math library does to it), plus the launch-angle grid, the element and the pipe nudged by one ulp."""
    variants = []
    try:
        for mode in NOISE_MODES:
            cport.set_trig_noise(mode)
            variants.append(cport.shoot_batch(xa, za, zf, alpha, geoms))
    finally:
        cport.set_trig_noise(0)
    base = cport.shoot_batch(xa, za, zf, alpha, geoms)
    for k in (1, -1):
        g_o = geoms.copy(); g_o[:, 1] = nudge(geoms[:, 1], k)  # an offset of exactly 0 with the element on the axis is the
        variants += [cport.shoot_batch(xa, za, zf, nudge(alpha, k), geoms),      # degenerate continuum (every ray returns
                     cport.shoot_batch(nudge(xa, k), za, zf, alpha, g_o)]        # through its own polyline vertex)
    spread = np.zeros_like(base)
    stable = np.ones(base.shape, dtype=bool)
    nb = np.isnan(base)
    for v in variants:
        nv = np.isnan(v)
        stable &= (nv == nb)
        with np.errstate(invalid="ignore"):
            spread = np.fmax(spread, np.where(nv | nb, 0.0, np.abs(v - base)))
    return base, spread, stable


def wrong_segment(got):
    """Self-test corruption: every 97th ray with a finite second lens point is moved onto the next polyline chord."""
    got = got.copy()
    xp, zp = got[0], got[1]                                   # lens_1 = the polyline of the alpha grid (main_rt.py:338, 390)
    n = xp.size
    moved = 0
    for r in range(0, n, 97):
        xq, zq, xi, zi = got[2, r], got[3, r], got[4, r], got[5, r]
        if not np.isfinite([xq, zq, xi, zi]).all() or xi == xq:
            continue
        m = (zi - zq) / (xi - xq); b = zq - m * xq
        j = int(np.searchsorted(xp, xi)) - 1                  # chord [j, j+1] holds the intersection
        j2 = j + 1
        if j < 0 or j2 + 1 >= n:
            continue
        ms = (zp[j2 + 1] - zp[j2]) / (xp[j2 + 1] - xp[j2]); bs = zp[j2] - ms * xp[j2]
        x = (bs - b) / (m - ms)
        got[4, r], got[5, r] = x, m * x + b
        moved += 1
    return got, moved


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    selftest = "--selftest" in sys.argv
    keep_going = "--keep-going" in sys.argv                    # count failing traces instead of stopping at the first (comparing builds)
    n_fail = 0
    trials = int(args[0]) if len(args) > 0 else 150
    rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 12345)
    worst_ratio, worst_abs, rays, decided, noted, hard_total, t0 = 0.0, 0.0, 0, 0, 0, 0, time.time()
    second_looks, second_passes = 0, 0                         # traces that failed the first rule / of those, that passed the second look
    for trial in range(trials):
        n = int(rng.choice([rng.integers(2, 20), rng.integers(20, 600), rng.integers(600, 4200), rng.integers(4200, 9000)]))
        if rng.random() < 0.5:
            alpha = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)
        else:
            alpha = np.sort(rng.uniform(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n))
        zf = np.full(n, D) + (rng.uniform(-1e-3, 1e-3, n) if rng.random() < 0.3 else 0.0)
        r = rng.uniform(0.005, 0.12, 3)
        off = rng.uniform(-0.02, 0.02, 3)
        kind = rng.integers(0, 4)
        if kind == 0: off[0] = 0.0
        if kind == 1: off[1] = r[1] * rng.choice([-1.0, 1.0])
        geoms = np.stack([r, off], axis=1)
        xa = np.concatenate([[0.0], rng.uniform(-0.02, 0.02, 2)])
        za = np.full(3, D)
        got_all = rtus.shoot_batch(xa, za, zf, alpha, geoms, params=rtus.Params(), want=("out8",))["out8"]
        base, spread, stable = oracle_with_spread(xa, za, zf, alpha, geoms)
        for gi in range(3):
            for t in range(3):
                got, o, sp, stb = got_all[gi, t], base[gi, t], spread[gi, t], stable[gi, t]
                if selftest:
                    got, _ = wrong_segment(got)
                ray_stable = stb.all(axis=0)                   # the oracle keeps this ray's NaN pattern under every nudge
                mism = (np.isnan(got) != np.isnan(o)).any(axis=0)
                n_unstable = int((~ray_stable).sum())
                hard = mism & ray_stable
                # A trace in which the oracle itself flips the NaN status of U rays under one-ulp nudges (a random grid with
                # near-duplicate vertices at offset 0 on the axis: every ray returns through its own vertex and the +-1e-9
                # bounds check next to the duplicate is decided by rounding) holds more such rays than twelve nudges expose:
                # the GPU may disagree on U/4 further rays there.  U = 0 (every other trace): no disagreement at all.
                if hard.sum() > n_unstable // 4 and n_unstable > 0:
                    # a degenerate trace (the oracle itself flips rays of it): the fourteen variants sample its rounding-decided
                    # rays thinly.  Second look with 48 more noise patterns, same rule: disagreements only on rays the reference's
                    # own arithmetic decides by rounding, plus a quarter of their number.
                    more = nan_stable_under(MORE_NOISE_MODES, xa[t:t + 1], za[t:t + 1], zf, alpha, geoms[gi:gi + 1])[0, 0]
                    second_looks += 1
                    first_rule = (int(hard.sum()), n_unstable)
                    ray_stable = ray_stable & more
                    n_unstable = int((~ray_stable).sum())
                    hard = mism & ray_stable
                    print(f"note: degenerate trace, second look (62 noise patterns): trial {trial} geom={geoms[gi]} xa={xa[t]}: "
                          f"{n_unstable} ray(s) flip in the oracle, {int(hard.sum())} disagreement(s) elsewhere "
                          f"(first rule, 14 patterns: {first_rule[0]} disagreement(s) against {first_rule[1]} flipping ray(s))")
                    second_passes += int(hard.sum() <= n_unstable // 4)
                if hard.sum() > n_unstable // 4:
                    bad = np.flatnonzero(hard)
                    print(f"NaN MASK MISMATCH trial {trial} n={n} geom={geoms[gi]} xa={xa[t]}: {hard.sum()} ray(s), first {bad[:5].tolist()}, "
                          f"whose NaN status the oracle keeps under every last-bit perturbation ({n_unstable} ray(s) of this trace do flip)")
                    if keep_going:
                        n_fail += 1
                        continue
                    sys.exit(3 if selftest else 1)
                decided += int((mism | ~ray_stable).sum())
                hard_total += int(hard.sum())
                m = ~np.isnan(o) & ~np.isnan(got)
                with np.errstate(invalid="ignore"):
                    d = np.where(m, np.abs(got - o), 0.0)
                allowed = np.where(ray_stable[None, :], ABS_TOL + K_SPREAD * sp, np.inf)   # value <-> NaN under a nudge: no bound
                ratio = d / allowed
                if (ratio > 1.0).any():
                    k, ray = np.unravel_index(np.argmax(ratio), ratio.shape)
                    print(f"VALUE MISMATCH trial {trial} n={n} geom={geoms[gi]} xa={xa[t]}: {rtus.KEYS[k]}[{ray}] gpu {got[k, ray]!r} "
                          f"oracle {o[k, ray]!r} |d| {d[k, ray]:.3e} > 1e-12 + 16 x spread {sp[k, ray]:.3e}; ray (oracle): {o[:, ray].tolist()}")
                    if keep_going:
                        n_fail += 1
                        continue
                    sys.exit(3 if selftest else 1)
                worst_ratio = max(worst_ratio, float(ratio.max()))
                worst_abs = max(worst_abs, float(d.max()))
                if d.max() > ABS_TOL:                          # an ill-conditioned ray inside its own spread: report it
                    k, ray = np.unravel_index(np.argmax(d), d.shape)
                    noted += 1
                    if noted <= 40:
                        print(f"note: |d| {d[k, ray]:.3e} m within 16 x spread {sp[k, ray]:.3e} m: trial {trial} n={n} geom={geoms[gi]} "
                              f"xa={xa[t]} {rtus.KEYS[k]}[{ray}]")
                rays += n
        if trial % 25 == 24:
            print(f"trial {trial + 1}/{trials}: {rays} rays checked, worst |d| {worst_abs:.2e} m, worst |d| / allowed {worst_ratio:.3f}, "
                  f"{time.time() - t0:.0f} s", flush=True)
    if keep_going:
        print(f"--keep-going: {n_fail} failing trace(s) of {trials * 9}")
    if selftest:
        print("SELFTEST FAILED: the off-by-one segment index was not caught")
        sys.exit(1)
    print(f"OK: {trials} trials, {rays} rays, worst |d| {worst_abs:.2e} m, worst |d| / (1e-12 + 16 spread) {worst_ratio:.3f}; "
          f"{noted} case(s) above 1e-12 m inside their own spread; {decided} rounding-decided ray(s) (NaN status moves with 1-2 ulp of "
          f"the inputs in the oracle itself, or next to such rays: {hard_total}); "
          f"SECOND LOOKS: {second_looks} trace(s) of {trials * 9} failed the first rule (14 noise patterns), {second_passes} of them passed with 62")


if __name__ == "__main__":
    main()

"""Fuzz of the forward trace (reference-compatible mode) against the C oracle: random ray counts (2 .. 9000: all
box-hierarchy depths, ragged leaves), pipe radii / offsets including offset = 0 (every ray retraces itself) and
|offset| = r (tangent pipes), off-centre elements, uniform and non-uniform launch-angle grids, per-ray landing depths.
Checks NaN masks (identical) and values of all eight outputs: scaled difference |d| / (1 m + 100 |value|) is reported
above 1e-12, counted above 1e-9 (at most ~1 ray in 10^6 may get there) and fatal above 1e-7.

    gpurun -- python scripts/fuzz_shoot.py [n_trials] [seed]
"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtus
from oracle import cport

D = float(np.float64(0.12156646438729327) + np.float64(0.08843353561270673))
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
worst, rays, noise_masks, ill, t0 = 0.0, 0, 0, 0, time.time()
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
    b = rtus.shoot_batch(xa, za, zf, alpha, geoms, params=rtus.Params(), want=("out8",))
    for gi in range(3):
        for t in range(3):
            o, _ = cport.shoot(xa[t], za[t], zf, alpha, geoms[gi, 0], geoms[gi, 1])
            got = b["out8"][gi, t]
            # offset = 0 with the element on the axis: every ray returns through its own polyline vertex, d_r is pure
            # rounding noise and which neighbouring segment "holds the sign change" (or fails the +-1e-9 bounds check
            # next to a near-duplicate vertex of a random grid) is decided by the last bit of tan/atan — ocml and glibc
            # legitimately disagree there (the oracle itself returns scattered NaNs).  Values must still agree.
            degenerate = geoms[gi, 1] == 0.0 and xa[t] == 0.0
            if not np.array_equal(np.isnan(got), np.isnan(o)):
                bad = np.argwhere(np.isnan(got) != np.isnan(o))
                if not degenerate or len(bad) > 0.05 * got.size:
                    print(f"NaN MASK MISMATCH trial {trial} n={n} geom={geoms[gi]} xa={xa[t]} first at {bad[:5].tolist()}")
                    sys.exit(1)
                noise_masks += len(bad)
            m = ~np.isnan(o) & ~np.isnan(got)
            # near-horizontal landing rays put target_x tens of metres away: 1-ulp libm differences (ocml vs glibc)
            # scale with the value, so the bound is 1e-12 m + 1e-10 |value| (a wrong segment would show as >= 1e-5)
            d = float(np.max(np.abs(got[m] - o[m]) / (1.0 + 100.0 * np.abs(o[m])))) if m.any() else 0.0
            worst = max(worst, d)
            if d > 1e-12:                                    # report (ill-conditioned rays) ...
                sc = np.where(m, np.abs(got - o) / (1.0 + 100.0 * np.abs(o)), 0.0)
                k, ray = np.unravel_index(np.argmax(sc), sc.shape)
                print(f"note: scaled diff {d:.3e} trial {trial} n={n} geom={geoms[gi]} xa={xa[t]}: {rtus.KEYS[k]}[{ray}] "
                      f"gpu {got[k, ray]!r} oracle {o[k, ray]!r}; ray's outputs (oracle): {o[:, ray].tolist()}")
            if d > 1e-9:                                     # a ray at the critical angle on top of a grazing chord: ~1 in 10^7
                ill += 1
            if d > 1e-7 or ill > 3 + 1e-6 * rays:            # ... fail on anything a wrong segment / branch would cause (>= 1e-5)
                print("VALUE MISMATCH")
                sys.exit(1)
            rays += n
    if trial % 25 == 24:
        print(f"trial {trial + 1}/{trials}: {rays} rays checked, worst scaled |dx| {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"OK: {trials} trials, {rays} rays, worst scaled |dx| {worst:.2e}; {ill} ray(s) above 1e-9; {noise_masks} noise-decided NaN flags in degenerate (offset 0, on-axis) cases")

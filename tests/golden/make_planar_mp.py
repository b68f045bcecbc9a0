"""Generates tests/golden/planar_mp.npz — 50-digit reference values for the planar-layer Fermat solves.

The reference repository has no planar interfaces, so these travel times cannot be pinned to
it.  What CAN be pinned is the arithmetic: this script solves Snell's law for the ray parameter p
(sum_i h_i p c_i / sqrt(1 - p^2 c_i^2) = X, bisection in mpmath at 50 significant digits — a formulation and a
number system independent of both the C oracle's long-double q-bisection and the GPU's Newton-in-q) and stores
T = sum_i h_i / (c_i sqrt(1 - p^2 c_i^2)) rounded once to float64.

    python tests/golden/make_planar_mp.py        (needs mpmath; about a minute)
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50


def layer_thickness(z_if, ze, zf):
    """Path thickness inside each layer for a source at depth ze and a target at depth zf > ze."""
    edges = [mp.mpf("-inf")] + [mp.mpf(float(z)) for z in z_if] + [mp.mpf("inf")]
    ze, zf = mp.mpf(float(ze)), mp.mpf(float(zf))
    return [max(min(edges[i + 1], zf) - max(edges[i], ze), mp.mpf(0)) for i in range(len(edges) - 1)]


def travel_time(z_if, c, xe, ze, xf, zf):
    if not zf > ze:
        return float("nan")
    h = layer_thickness(z_if, ze, zf)
    cc = [mp.mpf(float(v)) for v in c]
    X = abs(mp.mpf(float(xf)) - mp.mpf(float(xe)))
    used = [i for i in range(len(cc)) if h[i] > 0]
    cmax = max(cc[i] for i in used)
    if X == 0:
        return float(sum(h[i] / cc[i] for i in used))
    reach = lambda p: sum(h[i] * p * cc[i] / mp.sqrt(1 - (p * cc[i]) ** 2) for i in used)
    lo, hi = mp.mpf(0), 1 / cmax                      # reach(lo) = 0 < X < reach(hi^-) = inf
    for _ in range(200):                                # 2^-200: far below 50 digits
        mid = (lo + hi) / 2
        if mid == lo or mid == hi:
            break
        if reach(mid) < X:
            lo = mid
        else:
            hi = mid
    p = (lo + hi) / 2
    return float(sum(h[i] / (cc[i] * mp.sqrt(1 - (p * cc[i]) ** 2)) for i in used))


def case(name, z_if, c, xe, ze, xf, zf):
    z_if, c, xe, ze, xf, zf = (np.asarray(a, dtype=np.float64) for a in (z_if, c, xe, ze, xf, zf))
    tt = np.array([[travel_time(z_if, c, xe[e], ze[e], xf[f], zf[f]) for f in range(xf.size)] for e in range(xe.size)])
    print(name, tt.shape, "finite", int(np.isfinite(tt).sum()))
    return {f"{name}_z_if": z_if, f"{name}_c": c, f"{name}_xe": xe, f"{name}_ze": ze, f"{name}_xf": xf, f"{name}_zf": zf,
            f"{name}_tt": tt}


def main():
    rng = np.random.default_rng(2024)
    out = {}
    # BASELINE config 2 medium: 1 interface, slow under fast
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 9), np.linspace(0.025, 0.065, 5))
    out.update(case("cfg2", [0.020], [2330.0, 1483.0], (np.arange(0, 128, 21) - 63.5) * 0.6e-3, np.zeros(7), xs.ravel(), zs.ravel()))
    # config 3 medium: 2 interfaces, targets above / between / below them
    xs, zs = np.meshgrid(np.linspace(-0.03, 0.03, 7), np.array([0.004, 0.0099, 0.018, 0.0251, 0.06]))
    out.update(case("cfg3", [0.010, 0.025], [2330.0, 1483.0, 5900.0], (np.arange(0, 256, 51) - 127.5) * 0.3e-3, np.zeros(6),
                    xs.ravel(), zs.ravel()))
    # 9 layers, random speeds, offsets up to 100x the depth (near-critical rays)
    z_if = np.cumsum(rng.uniform(0.002, 0.01, 8))
    c = rng.uniform(1000.0, 6500.0, 9)
    out.update(case("deep", z_if, c, rng.uniform(-0.05, 0.05, 4), rng.uniform(-0.004, 0.0015, 4),
                    np.concatenate([rng.uniform(-0.4, 0.4, 24), [5.0, -5.0]]),
                    np.concatenate([rng.uniform(0.002, z_if[-1] + 0.02, 24), [0.03, z_if[-1] + 0.001]])))
    # elements inside deeper layers, targets in the element's own layer, target above the element (NaN)
    out.update(case("inner", [0.005, 0.012, 0.020], [1500.0, 3200.0, 1480.0, 5900.0], [-0.01, 0.0, 0.003, 0.02],
                    [0.0, 0.006, 0.0125, 0.0199], np.concatenate([rng.uniform(-0.05, 0.05, 20), [0.0, 0.003]]),
                    np.concatenate([rng.uniform(0.0005, 0.03, 20), [0.0061, 0.0126]])))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "planar_mp.npz"), **out)


if __name__ == "__main__":
    main()

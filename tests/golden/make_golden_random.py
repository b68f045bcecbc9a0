#!/usr/bin/env python3
"""More golden vectors from RUNNING the reference (see make_golden.py): shoot_rays on inputs its own drivers never use —
non-uniform launch-angle grids, a landing depth z_f that differs from ray to ray, random geometries and transmit positions.
The accelerated path takes arbitrary alpha / z_f arrays; these cases pin that generality to the reference itself.

Usage:  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_random.py
Output: tests/golden/random_cfg.npz  (inputs + the reference's 8 result arrays per case; cases the reference raises on are
recorded as such: `raises` = 1, no outputs)
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (imports the reference read-only)

M = G.M


def main():
    rng = np.random.default_rng(20261004)
    out = {}
    n_cases = 20
    for i in range(n_cases):
        n = int(rng.choice([64, 97, 181, 256, 300, 400]))
        r_outer = float(rng.uniform(0.012, 0.09))
        off = float(rng.uniform(-0.009, 0.009))
        x_tx = float(rng.uniform(-0.0189, 0.0189))
        G.set_globals(n, r_outer, off)
        amax = float(M.alpha_max)
        kind = i % 4
        if kind == 0:                                   # the drivers' own grid, other geometry / tx
            alpha = np.linspace(-amax, amax, n)
        elif kind == 1:                                 # non-uniform ascending grid
            alpha = np.sort(rng.uniform(-amax, amax, n))
        elif kind == 2:                                 # a narrow fan around a random direction
            c = rng.uniform(-0.5, 0.5)
            alpha = np.linspace(max(c - 0.2, -amax), min(c + 0.2, amax), n)
        else:                                           # descending grid (the polyline then runs right to left)
            alpha = np.linspace(amax, -amax, n)
        zf = np.ones(n, dtype=np.float64) * M.d
        if i % 3 == 1:
            zf = zf + rng.uniform(-0.004, 0.004, n)     # landing depth per ray
        if i % 5 == 4:
            zf = np.ones(n, dtype=np.float64) * (M.d + 0.01)
        cfg = np.asarray([n, r_outer, off, x_tx], dtype=np.float64)
        out[f"c{i:02d}_cfg"] = cfg
        out[f"c{i:02d}_alpha"] = alpha
        out[f"c{i:02d}_zf"] = zf
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = M.shoot_rays(np.float64(x_tx), M.d, zf, alpha, plot=False)
            out8 = np.stack([np.asarray(res[k], dtype=np.float64) for k in G.KEYS])
            out[f"c{i:02d}_out8"] = out8
            out[f"c{i:02d}_raises"] = np.asarray(0)
            print(i, n, kind, "ok, valid rays", int(np.isfinite(out8[6]).sum()))
        except Exception as e:                          # e.g. LinAlgError inside np.polyfit when a ray misses the pipe (Q6)
            out[f"c{i:02d}_raises"] = np.asarray(1)
            out[f"c{i:02d}_exc"] = np.asarray(type(e).__name__)
            print(i, n, kind, "reference raises", type(e).__name__)
    out["n_cases"] = np.asarray(n_cases)
    np.savez_compressed(os.path.join(G.OUT, "random_cfg.npz"), **out)


if __name__ == "__main__":
    main()

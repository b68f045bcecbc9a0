"""Scratch (GPU box): worst relative error of the tau-p tier vs the oracle on the irregular / coarse apertures of
tests/test_gpu_irregular_apertures.py (RTUS_LIB=... to look at another build)."""
import os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rtus
from oracle import cport
import test_gpu_irregular_apertures as T
for z_if, c in (([0.020], [2330.0, 1483.0]), ([0.010, 0.025], [1483.0, 5900.0, 2330.0])):
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, T.GRID[0]), np.linspace(0.004, 0.06, T.GRID[1]))
    xf, zf = xs.ravel(), zs.ravel()
    rng = np.random.default_rng(3)
    cases = list(T._apertures())
    cases.append(("coarse 5 mm", (np.arange(40) - 19.5) * 5e-3, np.zeros(40)))
    cases.append(("fine then coarse", np.concatenate([(np.arange(20) - 30) * 0.3e-3, 0.001 + np.arange(20) * 4e-3]), np.zeros(40)))
    cases.append(("random spacing", np.sort(rng.uniform(-0.03, 0.03, 40)), np.zeros(40)))
    for name, xe, ze in cases:
        fast = rtus.travel_time_layers(z_if, c, xe, ze, xf, zf, taup=True)
        ref = cport.tt_layers(z_if, c, xe, ze, xf, zf)
        ok = zf[None, :] > ze[:, None]
        rel = np.where(ok, np.abs(fast - ref) / ref, 0.0)
        r, f = np.unravel_index(np.argmax(rel), rel.shape)
        print(f"{len(c)} layers {name:18s}: worst rel {rel.max():.2e} (abs {abs(fast[r, f] - ref[r, f]):.2e} s) at row {r} xe {xe[r]:+.5f} target ({xf[f]:+.5f}, {zf[f]:.5f}); > 6e-11: {(rel > 6e-11).sum()}")

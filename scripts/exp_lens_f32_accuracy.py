"""Scratch (GPU box): fp32 curved-lens table against the fp64 table of the same inputs — max |dT| overall and per row class —
for the configs[3] geometry at a reduced grid, and a pitch / depth sweep that stresses the extrapolated starts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import rtus

p = rtus.Params()
for n_e, pitch, g, z0, z1 in ((1024, 0.3e-4, 256, 0.03, 0.07), (256, 1.5e-4, 256, 0.012, 0.03), (64, 6e-4, 300, 0.02, 0.2), (333, 1.1e-4, 128, 0.0, 0.05)):
    xe = (np.arange(n_e) - (n_e - 1) / 2) * pitch
    ze = np.full(n_e, p.d)
    xl, zl = np.meshgrid(np.linspace(-0.03, 0.03, g), np.linspace(z0, z1, g))
    t64 = rtus.travel_time_lens(xe, ze, xl.ravel(), zl.ravel(), params=p, dtype=np.float64)
    t32 = rtus.travel_time_lens(xe, ze, xl.ravel(), zl.ravel(), params=p, dtype=np.float32)
    d = np.abs(t32.astype(np.float64) - t64)
    ulp = np.spacing(t32.astype(np.float32)).astype(np.float64)
    print(f"n_e {n_e} pitch {pitch:g} grid {g}x{g} z {z0}-{z1}: max |dT| {np.nanmax(d):.3e} s, in fp32 ulps of T max {np.nanmax(d / ulp):.2f}, "
          f"mean {np.nanmean(d / ulp):.3f}; rows by position in block of 3: " +
          ", ".join(f"{np.nanmax(d[r::3]):.2e}" for r in range(3)), flush=True)

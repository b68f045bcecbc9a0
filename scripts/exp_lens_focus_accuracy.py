"""Scratch (GPU box): fp32 and fp64 curved-lens tables for targets AROUND THE LENS FOCUS (T flat in alpha: the minimiser is
ill-conditioned there, which is where rows that skip the Newton step could go wrong) against an fp64 table computed with the alpha
output (that path never skips a step); one library per process (RTUS_LIB)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import rtus
p = rtus.Params()
tag = os.environ.get("RTUS_LIB", "tree").split("/")[-1]
for n_e, pitch in ((1024, 3e-5), (256, 1.2e-4), (128, 3e-4)):
    xe = (np.arange(n_e) - (n_e - 1) / 2) * pitch
    ze = np.full(n_e, p.d)
    for x0, z0, half in ((0.0, 0.0, 2e-3), (0.0, 0.004, 4e-3), (0.003, 0.001, 1e-3)):
        xl, zl = np.meshgrid(np.linspace(x0 - half, x0 + half, 96), np.linspace(max(z0 - half, 1e-5), z0 + half, 96))
        ref, _ = rtus.travel_time_lens(xe, ze, xl.ravel(), zl.ravel(), params=p, dtype=np.float64, return_alpha=True)
        t64 = rtus.travel_time_lens(xe, ze, xl.ravel(), zl.ravel(), params=p, dtype=np.float64)
        t32 = rtus.travel_time_lens(xe, ze, xl.ravel(), zl.ravel(), params=p, dtype=np.float32)
        print(f"{tag} n_e {n_e} pitch {pitch:g} targets ({x0:g}, {z0:g}) +- {half:g}: fp64 table vs fp64 with alpha output max |dT| {np.nanmax(np.abs(t64 - ref)):.3e} s; "
              f"fp32 vs that: max {np.nanmax(np.abs(t32 - ref)):.3e} s, mean {np.nanmean(np.abs(t32 - ref)):.3e} s", flush=True)

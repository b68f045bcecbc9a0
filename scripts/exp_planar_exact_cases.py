"""Scratch (GPU box): the planar kernel on inputs a random fuzz never draws — targets and elements EXACTLY on interface depths, targets
exactly below elements (dx = 0) and exactly at element positions, a whole table above its aperture — on tables with >= 8 rows per
workgroup, both tiers, plain (as given) and sorted entries, against the long-double oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import rtus
from importlib import import_module
from oracle import cport
dev = import_module("ray-tracing-ultrasound_amd.device")
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
z_if, c = [0.010, 0.025], [1483.0, 5900.0, 2330.0]
xe = (np.arange(40) - 19.5) * 0.6e-3
bad = 0
def check(name, xe, ze, xf, zf):
    global bad
    assert dev.rows_per_block(xe.size, xf.size) >= 8
    ref = cport.tt_layers(z_if, c, xe, ze, xf, zf)
    for taup in (False, True):
        for entry in ("plain", "sorted"):
            tt = (dev.tt_layers_dev(z_if, c, t(xe), t(ze), t(xf), t(zf), taup=taup).cpu().numpy() if entry == "plain"
                  else rtus.travel_time_layers(z_if, c, xe, ze, xf, zf, taup=taup))
            same_nan = np.array_equal(np.isnan(tt), np.isnan(ref))
            m = ~np.isnan(ref)
            rel = float(np.max(np.abs(tt - ref)[m] / ref[m])) if m.any() and same_nan else float("nan")
            ok = same_nan and (not m.any() or rel < (6e-11 if taup else 2e-11))
            bad += not ok
            print(f"{name:44s} taup={taup!s:5s} {entry:6s}: NaN masks equal {same_nan}, finite {int(m.sum())}, worst rel {rel:.2e} {'' if ok else '  <-- FAIL'}")
xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 240), np.linspace(0.004, 0.06, 220))
# 1. targets exactly on the interfaces and at the aperture's depth
zs1 = zs.copy(); zs1[50] = 0.010; zs1[51] = 0.025; zs1[52] = 0.0; zs1[53] = np.nextafter(0.010, 1); zs1[54] = np.nextafter(0.025, 0)
check("targets exactly on interfaces / at z = 0", xe, np.zeros(40), xs.ravel(), zs1.ravel())
# 2. elements exactly on an interface, and inside the second layer
check("elements exactly on the first interface", xe, np.full(40, 0.010), xs.ravel(), zs.ravel())
check("elements at depths 0 / 0.010 / 0.0175 by blocks", xe, np.repeat([0.0, 0.010, 0.0175, 0.010, 0.0], 8), xs.ravel(), zs.ravel())
# 3. targets exactly below elements / exactly at element x (dx = 0 on whole columns)
xs3 = xs.copy(); xs3[:, :40] = xe[None, :]
check("target columns at the elements' own x", xe, np.zeros(40), xs3.ravel(), zs.ravel())
# 4. the whole table above the aperture, and half of it
check("all targets above the aperture", xe, np.full(40, 0.07), xs.ravel(), zs.ravel())
check("aperture at mid depth (half the targets above it)", xe, np.full(40, 0.03), xs.ravel(), zs.ravel())
print("FAILURES:", bad)
sys.exit(1 if bad else 0)

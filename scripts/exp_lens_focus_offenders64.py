"""Scratch (GPU box): where the fp64 curved-lens table (rows may skip the Newton step) differs from the fp64 table computed with the
alpha output (never skips), around the lens focus at the configs[3] pitch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import rtus
p = rtus.Params()
n_e, pitch = 1024, 3e-5
xe = (np.arange(n_e) - (n_e - 1) / 2) * pitch
ze = np.full(n_e, p.d)
xl, zl = np.meshgrid(np.linspace(-2e-3, 2e-3, 96), np.linspace(1e-5, 2e-3, 96))
xf, zf = xl.ravel(), zl.ravel()
ref, aref = rtus.travel_time_lens(xe, ze, xf, zf, params=p, dtype=np.float64, return_alpha=True)
t = rtus.travel_time_lens(xe, ze, xf, zf, params=p, dtype=np.float64)
d = np.abs(t - ref)
print("max %.3e; entries > 1e-18: %d, > 1e-15: %d, > 1e-13: %d of %d" % (d.max(), (d > 1e-18).sum(), (d > 1e-15).sum(), (d > 1e-13).sum(), d.size))
for k in np.argsort(d.ravel())[::-1][:14]:
    e, f = divmod(k, xf.size)
    print("elem %4d (in block %3d) x_e %+.6f | target (%+.6f, %.6f) | T_ref %.12e dT %+.2e | alpha_ref %+.9f (max %.9f) | neighbours' alpha %+.9f %+.9f" %
          (e, e % 64, xe[e], xf[f], zf[f], ref[e, f], t[e, f] - ref[e, f], aref[e, f], rtus.ALPHA_MAX, aref[max(e - 1, 0), f], aref[min(e + 1, n_e - 1), f]))

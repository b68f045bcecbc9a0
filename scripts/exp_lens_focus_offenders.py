"""Scratch (GPU box): the worst fp32 entries of a curved-lens table around the lens focus — where, and what the fp32 solver did."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import rtus
p = rtus.Params()
n_e, pitch = 256, 1.2e-4
xe = (np.arange(n_e) - (n_e - 1) / 2) * pitch
ze = np.full(n_e, p.d)
xl, zl = np.meshgrid(np.linspace(-2e-3, 2e-3, 96), np.linspace(1e-5, 2e-3, 96))
xf, zf = xl.ravel(), zl.ravel()
ref, aref = rtus.travel_time_lens(xe, ze, xf, zf, params=p, dtype=np.float64, return_alpha=True)
t32, a32 = rtus.travel_time_lens(xe, ze, xf, zf, params=p, dtype=np.float32, return_alpha=True)
t32n = rtus.travel_time_lens(xe, ze, xf, zf, params=p, dtype=np.float32)
d = np.abs(t32.astype(np.float64) - ref)
dn = np.abs(t32n.astype(np.float64) - ref)
print("fp32 with alpha output: max %.3e, entries > 1e-9: %d of %d; without: max %.3e, entries > 1e-9: %d" % (d.max(), (d > 1e-9).sum(), d.size, dn.max(), (dn > 1e-9).sum()))
idx = np.argsort(d.ravel())[::-1][:12]
for k in idx:
    e, f = divmod(k, xf.size)
    print("elem %3d x_e %+.5f | target (%+.6f, %.6f) | T64 %.9e T32 %.9e dT %.2e | alpha64 %+.6f alpha32 %+.6f (alpha_max %.6f)" %
          (e, xe[e], xf[f], zf[f], ref[e, f], t32[e, f], d[e, f], aref[e, f], a32[e, f], rtus.ALPHA_MAX))
# is the fp64 value the global minimum?  brute force over alpha for the worst entry
from oracle import cport
e, f = divmod(idx[0], xf.size)
al = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, 200001)
T = cport.lens_time(al, xe[e], ze[e], xf[f], zf[f]) if hasattr(cport, "lens_time") else None
if T is not None:
    j = np.argmin(T)
    print("brute force: min T %.9e at alpha %+.6f" % (T[j], al[j]))

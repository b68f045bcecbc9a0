"""Scratch (GPU box): where the ~53 us of one drop-in shoot_rays(N = 905) call go — Python wrapper, C host twin (uploads, two
kernels, download, sync), device-resident call + sync."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from importlib import import_module
import numpy as np, torch
import rtus
dev_api = import_module("ray-tracing-ultrasound_amd.device")
p = rtus.Params(r_outer=0.037, pipe_offset=0.0038)
n = 905
alpha = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n); zf = np.full(n, p.d)
def bench(fn, k=2000):
    for _ in range(50): fn()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(k): fn()
        best = min(best, (time.perf_counter() - t0) / k)
    return best * 1e6
print("rtus.shoot_rays            %.1f us" % bench(lambda: rtus.shoot_rays(0.0, p.d, zf, alpha, plot=False, params=p)))
print("rtus.shoot_batch(out8)     %.1f us" % bench(lambda: rtus.shoot_batch([0.0], [p.d], zf, alpha, params=p)))
L = rtus.lib(); lens = p.lens()
g = np.array([[0.037, 0.0038]]); xa = np.array([0.0]); za = np.array([p.d]); out8 = np.empty((8, n))
ptr = lambda a: a.ctypes.data
args = (C.byref(lens), ptr(g), 1, ptr(xa), ptr(za), 1, ptr(alpha), ptr(zf), n, ptr(out8), None, None, None, None, 0, 0)
print("ctypes rtus_shoot (host)   %.1f us" % bench(lambda: L.rtus_shoot(*args)))
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
plan = dev_api.ShootPlan(1, 1, n, want=("out8",), params=p)
a = [t64(g), t64(xa), t64(za), t64(alpha), t64(zf)]
def dev():
    plan.run(*a); torch.cuda.synchronize()
print("device-resident + sync     %.1f us" % bench(dev))
def dev_kept():
    plan.run(*a, polyline_ready=True); torch.cuda.synchronize()
print("  ... polyline kept        %.1f us" % bench(dev_kept))
h = torch.empty((8, n), dtype=torch.float64).pin_memory()
def dev_d2h():
    o = plan.run(*a); h.copy_(o["out8"].view(8, n), non_blocking=True); torch.cuda.synchronize()
print("  ... + D2H of the result  %.1f us" % bench(dev_d2h))

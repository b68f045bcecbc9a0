"""Forward-trace kernel time (reference geometry, 1024 tx x 8192 rays, compat + fast) for the working tree and experiment
builds: python scripts/ab_shoot_variants.py [variant ...]   (each run is its own process: RTUS_LIB is read at import)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os; sys.path.insert(0, %r)
import numpy as np, torch, rtus
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
d = rtus.Params().d
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
n = 8192
xa = (np.arange(1024) - 511.5) * 0.3e-3
a = (t64([[0.037, 0.0038]]), t64(xa), t64(np.full(1024, d)), t64(np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)), t64(np.full(n, d)))
for fast in (False, True):
    plan = dev_api.ShootPlan(1, 1024, n, want=("tof", "land_x"), params=rtus.Params(), fast=fast)
    for _ in range(3): plan.run(*a)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): plan.run(*a)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"  fast={int(fast)}: {best*1e3:7.1f} us per 8.39 M rays = {8.388608/best:6.2f} G rays/s   checksum {float(plan.out['tof'].nan_to_num().sum()):.12e}")
''' % ROOT
for v in [None] + sys.argv[1:]:
    env = dict(os.environ)
    if v:
        env["RTUS_LIB"] = os.path.join(ROOT, "variants", f"librtus_{v}.so")
    print("==", v or "working tree", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)

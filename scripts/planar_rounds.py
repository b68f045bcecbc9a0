"""How the planar kernel's time scales with the number of workgroup rounds (RTUS_EB=64 forced): n_e = 256 elements,
n_f = k x 65536 targets -> k x 1024 workgroups of 4 waves = k/2 rounds of 8 waves per SIMD."""
import os
import sys

import numpy as np
import torch

os.environ["RTUS_EB"] = "64"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from importlib import import_module  # noqa: E402

dev_api = import_module("ray-tracing-ultrasound_amd.device")
dev = torch.device("cuda", 0)
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
W = bench.planar_inputs("cfg3_planar", 0, 1)
xe, ze = t64(W["xe"]), t64(W["ze"])
for k in (1, 2, 3, 4, 6, 8, 16):
    n_f = k * 65536
    xf, zf = t64(np.resize(W["xf"], n_f)), t64(np.resize(W["zf"], n_f))
    out = torch.empty((W["n_e"], n_f), dtype=torch.float64, device=dev)
    plan = dev_api.LayersPlan(W["z_if"], W["c"], xe, ze, xf, zf, out=out)
    for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    best = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20 * 1e3)
    us = float(np.median(best))
    print(f"n_f = {n_f:8d}  workgroups {k * 1024:6d}  rounds of 8 waves/SIMD {k / 2:4.1f}   {us:8.1f} us   {us / k:6.1f} us per 1024 workgroups   {W['n_e'] * n_f / us / 1e6:8.1f} k Mrays/s")

"""Scratch timing (GPU box): the reference's loop body as two calls (rtus_shoot_dev + rtus_match_dev) against the fused entry
(rtus_sweep_dev), at the reference's sweep size and at scale; HIP events around graph replays of 20 passes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from importlib import import_module
import numpy as np
import torch
import rtus
import bench

dev_api = import_module("ray-tracing-ultrasound_amd.device")
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")


def timed(fn, k=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(k):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    best = 1e9
    for _ in range(reps):
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / k)
    return best * 1e3


for kind in ("ref_sweep", "ref_scale"):
    R = bench.ref_inputs(kind)
    G, T, N, E = R["geoms"].shape[0], R["xa"].size, R["n"], R["x_rx"].size
    a = [t64(R[k]) for k in ("geoms", "xa", "za", "alpha", "zf")]
    x_rx = t64(R["x_rx"])
    for fast in (False, True):
        plan = dev_api.ShootPlan(G, T, N, want=("tof", "land_x"), params=rtus.Params(), fast=fast)
        box = [None]
        def two():
            o = plan.run(*a)
            box[0] = dev_api.match_dev(o["land_x"].view(G * T, N), o["tof"].view(G * T, N), x_rx, out=box[0])
        def trace_only():
            plan.run(*a)
        f1 = dev_api.SweepPlan(G, T, N, E, want=("tof", "land_x"), params=rtus.Params(), fast=fast)
        f2 = dev_api.SweepPlan(G, T, N, E, params=rtus.Params(), fast=fast)
        k = 20 if kind == "ref_sweep" else 5
        print(kind, "fast" if fast else "compat", "trace only %.1f us | two calls %.1f us | fused (+tof, land_x) %.1f us | fused hits only %.1f us | fused, polyline kept %.1f us"
              % (timed(trace_only, k), timed(two, k), timed(lambda: f1.run(*a, x_rx), k), timed(lambda: f2.run(*a, x_rx), k),
                 timed(lambda: f2.run(*a, x_rx, polyline_ready=True), k)), flush=True)
        del plan, f1, f2

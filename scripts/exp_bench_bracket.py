"""Scratch (GPU box): where the wall-clock overhead of bench.py's timed bracket (K = 20: 70-110 us beside 2.7 ms of kernels) comes from.
The same hipGraph of 20 headline launches, replayed under variants of the bracket."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import rtus, bench
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
W = bench.workload_inputs("cfg3_planar", 0, 1)
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
out = torch.empty((W["n_e"], W["n_f"]), dtype=torch.float64, device="cuda")
plan = dev_api.LayersPlan(W["z_if"], W["c"], t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"]), out=out, taup=True)
K = 20
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(K): plan.run()
torch.cuda.current_stream().wait_stream(side)
for _ in range(110): g.replay()
torch.cuda.synchronize()
def bracket(mode):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode != "noev": ev0.record()
    g.replay()
    if mode != "noev": ev1.record()
    if mode == "spin":
        while not ev1.query(): pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt * 1e3 / K, (ev0.elapsed_time(ev1) / K if mode != "noev" else float("nan"))
first = [bracket("sync") for _ in range(4)]
print("the first four brackets of the process (wall, events) ms per step:", [(round(a, 5), round(b, 5)) for a, b in first], "-> overhead us per region:", [round((a - b) * K * 1e3, 1) for a, b in first])
for mode in ("sync", "spin", "noev", "sync", "spin", "noev"):
    r = [bracket(mode) for _ in range(15)]
    w = np.median([a for a, _ in r]); e = np.median([b for _, b in r])
    print(f"{mode:5s}: wall {w:.5f} ms per step, events {e:.5f}, bracket overhead {(w - e) * K * 1e3 if e == e else float('nan'):.1f} us per region")

"""configs[2] through the plain and the sorted entry, ordered and shuffled aperture (a few launches each; run under rocprofv3
--kernel-trace for the per-kernel split)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rtus, bench
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
W = bench.planar_inputs("cfg3_planar", 0, 1)
xf, zf = t64(W["xf"]), t64(W["zf"])
out = torch.empty((W["n_e"], W["n_f"]), dtype=torch.float64, device="cuda")
perm = np.random.default_rng(3).permutation(W["n_e"])
ws = torch.empty(int(rtus.lib().rtus_tt_layers_sort_workspace_bytes(W["n_e"])), dtype=torch.uint8, device="cuda")
for name, xe, ze in (("ordered", W["xe"], W["ze"]), ("shuffled", W["xe"][perm], W["ze"][perm])):
    xe, ze = t64(xe), t64(ze)
    for entry, fn in (("plain ", lambda: dev_api.tt_layers_dev(W["z_if"], W["c"], xe, ze, xf, zf, out=out, taup=True)),
                      ("sorted", lambda: dev_api.tt_layers_sorted_dev(W["z_if"], W["c"], xe, ze, xf, zf, out=out, ws=ws, taup=True))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print(f"{name:9s} {entry}: {best*1e3:7.1f} us per table")

"""Scratch (GPU box): BASELINE configs[4] whole table (2048 x 2048, 3 layers) and configs[1] (128 x 128^2, 1 interface) timed with graph
replays of 200 launches, one library per process (RTUS_LIB)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from importlib import import_module
import numpy as np, torch
import rtus, bench
dev_api = import_module("ray-tracing-ultrasound_amd.device")
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
for name, W in (("cfg5", bench.fmc_inputs(0, 1)), ("cfg2", bench.planar_inputs("cfg2_planar", 0, 1))):
    out = torch.empty((W["n_e"], W["n_f"]), dtype=torch.float64, device="cuda")
    a = [t64(W[k]) for k in ("xe", "ze", "xf", "zf")]
    for taup in (False, True):
        plan = dev_api.LayersPlan(W["z_if"], W["c"], *a, out=out, taup=taup)
        for _ in range(3): plan.run()
        torch.cuda.synchronize()
        timed = bench.Timed(torch, lambda s: plan.run(), 200)
        best = min(timed.run(lambda: None)[1] for _ in range(3))
        print(os.environ.get("RTUS_LIB", "tree").split("/")[-1], name, "taup" if taup else "accurate", "%.2f us" % (best * 1e3), flush=True)

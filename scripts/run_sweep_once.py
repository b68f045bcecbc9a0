"""One form of the reference's sweep per run, for rocprofv3 (scripts/profile_round.sh): `two` = rtus_shoot_dev + rtus_match_dev,
`fused` = rtus_sweep_dev, `kept` = rtus_sweep_dev with the polyline kept (RTUS_POLYLINE_READY); 50 eager passes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import numpy as np
import torch
import rtus
import bench

dev_api = import_module("ray-tracing-ultrasound_amd.device")
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
form = sys.argv[1] if len(sys.argv) > 1 else "fused"
R = bench.ref_inputs("ref_sweep")
G, T, N, E = R["geoms"].shape[0], R["xa"].size, R["n"], R["x_rx"].size
a = [t64(R[k]) for k in ("geoms", "xa", "za", "alpha", "zf")]
x_rx = t64(R["x_rx"])
if form == "two":
    plan = dev_api.ShootPlan(G, T, N, want=("tof", "land_x"), params=rtus.Params())
    box = [None]
    def one():
        o = plan.run(*a)
        box[0] = dev_api.match_dev(o["land_x"].view(G * T, N), o["tof"].view(G * T, N), x_rx, out=box[0])
else:
    plan = dev_api.SweepPlan(G, T, N, E, params=rtus.Params())
    one = lambda: plan.run(*a, x_rx, polyline_ready=(form == "kept"))
    plan.run(*a, x_rx)
for _ in range(3):
    one()
torch.cuda.synchronize()
for _ in range(50):
    one()
torch.cuda.synchronize()
print(form, "done")

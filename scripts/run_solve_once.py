"""A few device-resident solve passes (rtus_solve_dev) for rocprofv3: `sweep` (210 geometries x 65 rx, N = 905) or
`scale` (16 geometries x 1024 tx x 65 rx)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rtus
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
which = sys.argv[1] if len(sys.argv) > 1 else "sweep"
fast = len(sys.argv) > 2 and sys.argv[2] == "fast"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
d = rtus.Params().d
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
n = 905
alpha = t64(np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n))
x_rx = t64(rtus.reference_elements())
if which == "sweep":
    geoms, xa = np.array([[r * 1e-2, o * 1e-3] for r in range(1, 11) for o in range(-10, 11)]), np.array([0.0])
else:
    geoms, xa = np.array([[0.02 + 0.005 * i, 0.0004 * (i - 7.5)] for i in range(16)]), (np.arange(1024) - 511.5) * 0.3e-4
plan = dev_api.SolvePlan(len(geoms), len(xa), n, 65, params=rtus.Params(), fast=fast)
a = (t64(geoms), t64(xa), t64(np.full(len(xa), d)), alpha, x_rx)
for _ in range(reps):
    plan.run(*a)
torch.cuda.synchronize()
print("done", which, fast, reps)

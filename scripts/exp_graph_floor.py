"""hipGraph replay of K planar-kernel launches: time per launch vs problem size (where is the fixed cost?)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rtus
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
dev = torch.device("cuda")
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
K = 200
def run(n_e, g):
    xe = (np.arange(n_e) - (n_e - 1) / 2) * 0.6e-3
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, g), np.linspace(0.026, 0.065, g))
    out = torch.empty((n_e, g * g), dtype=torch.float64, device=dev)
    plan = dev_api.LayersPlan([0.02], [2330., 1483.], t64(xe), t64(np.zeros(n_e)), t64(xs.ravel()), t64(zs.ravel()), out=out)
    for _ in range(5): plan.run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(K): plan.run()
    torch.cuda.current_stream().wait_stream(side)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / K)
    return best * 1e3
for n_e, g in ((1, 8), (8, 128), (32, 128), (64, 128), (128, 128), (256, 128), (512, 128), (128, 256)):
    print(f"n_e={n_e:4d} grid={g}x{g}  {run(n_e, g):8.2f} us/launch", flush=True)

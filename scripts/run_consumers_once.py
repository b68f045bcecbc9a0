"""One launch each of the travel-time-table consumers (focal laws over the configs[2] table, TFM 64 x 64 x 2048 onto a
256 x 256 image) for rocprofv3 passes (scripts/profile_round.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
dev = torch.device("cuda")
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
W = bench.planar_inputs("cfg3_planar", 0, 1)
tt3 = dev_api.tt_layers_dev(W["z_if"], W["c"], t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"]))
out = torch.empty_like(tt3)
for _ in range(3):
    dev_api.focal_delays_dev(tt3, out=out)
n_el, n_t, fs = 64, 2048, 50e6
xe = (np.arange(n_el) - (n_el - 1) / 2.0) * 0.6e-3
xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 256), np.linspace(0.025, 0.045, 256))
tt = dev_api.tt_layers_dev([0.020], [2330.0, 1483.0], t64(xe), t64(np.zeros(n_el)), t64(xs.ravel()), t64(zs.ravel()))
fmc = torch.randn((n_el, n_el, n_t), dtype=torch.float32, device=dev)
img = torch.empty(xs.size, dtype=torch.float32, device=dev)
for _ in range(3):
    dev_api.tfm_dev(fmc, fs, tt, out=img)
torch.cuda.synchronize()

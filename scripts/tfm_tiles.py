"""TFM delay-and-sum: does the ORDER of the focal points matter?  Lanes = consecutive focal points; a wave gathers, per
(tx, rx) pair, the samples its 64 focal points point at.  Row-major order (64 neighbours along x) against 8 x 8 tiles and
16 x 4 / 4 x 16 strips of the same 256 x 256 image (a permutation of the focal list: the kernel is unchanged)."""
import os
import sys
from importlib import import_module

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev_api = import_module("ray-tracing-ultrasound_amd.device")
dev = torch.device("cuda", 0)
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
n_el, n_t, fs, g = 64, 2048, 50e6, 256
xe = (np.arange(n_el) - (n_el - 1) / 2.0) * 0.6e-3
xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, g), np.linspace(0.025, 0.045, g))
fmc = torch.randn((n_el, n_el, n_t), dtype=torch.float32, device=dev)
ref = None
for name, (tz, tx) in (("row-major 1 x 64", (1, 64)), ("tiles 8 x 8", (8, 8)), ("strips 4 z x 16 x", (4, 16)), ("strips 16 z x 4 x", (16, 4)),
                       ("strips 2 z x 32 x", (2, 32))):
    idx = np.arange(g * g).reshape(g // tz, tz, g // tx, tx).transpose(0, 2, 1, 3).ravel()      # focal list in tile order
    tt = dev_api.tt_layers_dev([0.020], [2330.0, 1483.0], t64(xe), t64(np.zeros(n_el)), t64(xs.ravel()[idx]), t64(zs.ravel()[idx]))
    img = torch.empty(g * g, dtype=torch.float32, device=dev)
    ms = bench._event_ms(torch, lambda: dev_api.tfm_dev(fmc, fs, tt, out=img), 10)
    full = torch.empty_like(img)
    full[torch.as_tensor(idx, device=dev)] = img
    if ref is None:
        ref = full.clone()
    print(f"{name:20s} {ms * 1e3:8.1f} us   {n_el * n_el * g * g / (ms * 1e-3) / 1e9:7.1f} G pair-samples/s   max |image - row-major image| "
          f"{float((full - ref).abs().max()):.2e} (image max {float(ref.abs().max()):.2e})")

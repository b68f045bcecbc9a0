"""Scratch (GPU box): one library per process (RTUS_LIB), the configs[3] shard (128 x 1024^2, fp32) and the whole table timed with
HIP events around 10 back-to-back launches; run alternately for the working tree and variants/librtus_prev.so."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from importlib import import_module
import numpy as np, torch
import rtus
dev_api = import_module("ray-tracing-ultrasound_amd.device")
t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")
p = rtus.Params()
g = 1024
xl, zl = np.meshgrid(np.linspace(-0.004, 0.004, g), np.linspace(0.03, 0.07, g))
xf, zf = t32(xl.ravel()), t32(zl.ravel())
for n_e in (128, 1024):
    xe = t32((np.arange(1024) - 511.5) * 0.3e-4)[:n_e].contiguous()
    ze = t32(np.full(n_e, p.d))
    out = torch.empty((n_e, g * g), dtype=torch.float32, device="cuda")
    run = lambda: dev_api.tt_lens_rows_dev(xe, ze, xf, zf, out, params=p, row0=0, n_rows_total=1024)
    for _ in range(3): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(os.environ.get("RTUS_LIB", "tree"), n_e, "rows: %.4f ms" % best, "checksum", float(out.double().sum()), flush=True)

"""Scratch (GPU box, variants/librtus_count.so): how many triples of the fp32 lens kernel run their two T-only rows."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from importlib import import_module
import numpy as np, torch
import rtus
dev_api = import_module("ray-tracing-ultrasound_amd.device")
t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")
p = rtus.Params()
g = 512
xl, zl = np.meshgrid(np.linspace(-0.004, 0.004, g), np.linspace(0.03, 0.07, g))
xf, zf = t32(xl.ravel()), t32(zl.ravel())
n_e = 128
xe = t32((np.arange(1024) - 511.5) * 0.3e-4)[:n_e].contiguous()
ze = t32(np.full(n_e, p.d))
out = torch.empty((n_e, g * g), dtype=torch.float32, device="cuda")
L = rtus.lib()
buf = (C.c_ulonglong * 8)()
L.rtus_dbg_read_lens(buf)
dev_api.tt_lens_rows_dev(xe, ze, xf, zf, out, params=p, row0=0, n_rows_total=1024)
L.rtus_dbg_read_lens(buf)
v = list(buf)
print("triples", v[0], "with T-only rows", v[1], "| lite attempts", v[2], "lite ok", v[3], "| full iterations", v[4], "of which because some lane has no usable g1", v[5], "| trips of the full iteration", v[6], "| lite steps left because a lane looks flat", v[7])

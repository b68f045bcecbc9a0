"""Scratch (GPU box, variants/librtus_count.so): distribution of the cubic predictor's miss |dq| / q over the four-history rows of
BASELINE configs[2] (and a coarser aperture)."""
import os, sys, ctypes as C, struct
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from importlib import import_module
import numpy as np, torch
import rtus, bench
dev_api = import_module("ray-tracing-ultrasound_amd.device")
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
L = rtus.lib()
buf = (C.c_ulonglong * 8)()
for name in ("cfg3_planar", "cfg2_planar"):
    W = bench.planar_inputs(name, 0, 1)
    out = torch.empty((W["n_e"], W["n_f"]), dtype=torch.float64, device="cuda")
    a = [t64(W[k]) for k in ("xe", "ze", "xf", "zf")]
    plan = dev_api.LayersPlan(W["z_if"], W["c"], *a, out=out, taup=True)
    L.rtus_dbg_read_planar(buf)
    plan.run(); torch.cuda.synchronize()
    L.rtus_dbg_read_planar(buf)
    v = list(buf)
    mx = struct.unpack("f", struct.pack("I", v[7] & 0xffffffff))[0]
    print(name, "four-history solves", v[0], "| miss > 3e-5:", v[1], "| WAVES (of %d) with a lane > 3e-5:" % (v[0] // 64), v[2], "> 5e-5:", v[3], "> 1e-4:", v[4], "| groups of four:", v[5], "held:", v[6], "| max %.3e" % mx)

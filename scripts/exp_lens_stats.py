"""Scratch (GPU box): how the lens kernel solved its rows (rtus_tt_lens[_f32]_stats_dev) on a few apertures; RTUS_LIB picks the build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import rtus
from importlib import import_module
dev = import_module("ray-tracing-ultrasound_amd.device")
p = rtus.Params()
xs, zs = np.meshgrid(np.linspace(-0.004, 0.004, 240), np.linspace(0.03, 0.07, 220))
for dt, n_e, pitch in ((torch.float64, 40, 0.5e-3), (torch.float32, 40, 0.5e-3), (torch.float64, 48, 0.6e-3), (torch.float32, 256, 0.3e-4), (torch.float64, 256, 0.3e-4), (torch.float32, 128, 3e-4)):
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    xe = (np.arange(n_e) - (n_e - 1) / 2) * pitch
    out = torch.empty((n_e, xs.size), dtype=dt, device="cuda")
    _, st = dev.tt_lens_stats_dev(t(xe), t(np.full(n_e, p.d)), t(xs.ravel()), t(zs.ravel()), out, params=p)
    w = st["wave_elements"]
    print(os.environ.get("RTUS_LIB", "tree").split("/")[-1], str(dt)[6:], n_e, pitch, "rows/wg", dev.rows_per_block(n_e, xs.size, dt),
          {k: round(v / w, 3) for k, v in st.items() if k != "wave_elements"}, "evaluations per row", round((st["t_only"] + st["one_evaluation"] + st["iteration_evaluations"]) / w, 3))

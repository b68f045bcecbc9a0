"""Curved-lens Fermat kernel: time per launch, fp64 and fp32 (BASELINE config 4 shape, 128 rows x 1024^2 targets)."""
import sys, os, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rtus
dev = torch.device("cuda")
n_e = 128
xe = (np.arange(n_e) - (n_e - 1) / 2.0) * 0.3e-4
xs, zs = np.meshgrid(np.linspace(-0.004, 0.004, 1024), np.linspace(0.03, 0.07, 1024))
lens = rtus.Params().lens()
for dt, fn in ((torch.float64, rtus.lib().rtus_tt_lens_dev), (torch.float32, rtus.lib().rtus_tt_lens_f32_dev)):
    npdt = np.float64 if dt == torch.float64 else np.float32
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=npdt), device=dev)
    a = [t(xe), t(np.full(n_e, rtus.Params().d)), t(xs.ravel()), t(zs.ravel())]
    out = torch.empty((n_e, xs.size), dtype=dt, device=dev)
    def run():
        st = fn(C.byref(lens), -rtus.ALPHA_MAX, rtus.ALPHA_MAX, a[0].data_ptr(), a[1].data_ptr(), n_e, a[2].data_ptr(),
                a[3].data_ptr(), xs.size, out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        assert st == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{dt}: {ms:8.3f} ms/launch  {n_e * xs.size / ms / 1e3:10.1f} Mrays/s  finite {float(torch.isfinite(out).float().mean()):.3f}")

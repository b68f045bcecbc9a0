"""Device-resident root-finding solve (rtus_solve_dev): time per pass by HIP events and under hipGraph replay, at the
reference sweep's size (210 geometries x 1 tx x 65 rx, N = 905) and scaled (16 geometries x 1024 tx x 65 rx)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rtus
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
d = rtus.Params().d
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
n = 905
alpha = t64(np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n))
x_rx = t64(rtus.reference_elements())
cases = {
    "sweep": (np.array([[r * 1e-2, o * 1e-3] for r in range(1, 11) for o in range(-10, 11)]), np.array([0.0])),
    "scale": (np.array([[0.02 + 0.005 * i, 0.0004 * (i - 7.5)] for i in range(16)]), (np.arange(1024) - 511.5) * 0.3e-4),
}
MODES = [m for m in sys.argv[1:]] or ["default"]
for fast, mode in [(f, m) for f in (False, True) for m in MODES]:
    for name, (geoms, xa) in cases.items():
        G, T = len(geoms), len(xa)
        plan = dev_api.SolvePlan(G, T, n, 65, params=rtus.Params(), fast=fast, all_roots=True, one_lane=mode == "one_lane",
                                 three_launches=mode == "three_launches")
        name = f"{name}/{mode}"
        a = (t64(geoms), t64(xa), t64(np.full(T, d)), alpha, x_rx)
        for _ in range(3):
            o = plan.run(*a)
        torch.cuda.synchronize()
        K = 50 if name == "sweep" else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            plan.run(*a)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(K):
                    plan.run(*a)
        torch.cuda.current_stream().wait_stream(side)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        msg = e0.elapsed_time(e1) / K
        nr = o["n_roots"].cpu().numpy()
        el = G * T * 65
        # the same with the polyline kept between passes (RTUS_POLYLINE_READY)
        g2 = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g2, stream=side):
                for _ in range(K):
                    plan.run(*a, polyline_ready=True)
        torch.cuda.current_stream().wait_stream(side)
        g2.replay(); torch.cuda.synchronize()
        e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
        msr = e0.elapsed_time(e1) / K
        o2 = {k: v.clone() for k, v in plan.out.items()}
        plan.run(*a); torch.cuda.synchronize()
        same = all(torch.equal(o2[k].view(torch.uint8), plan.out[k].view(torch.uint8)) for k in o2)
        print(f"{name:6s} fast={int(fast)}: polyline kept {msr*1e3:8.1f} us/pass -> {el/msr/1e3:8.1f} M/s, same bits {same}")
        print(f"{name:6s} fast={int(fast)}: eager {ms*1e3:8.1f} us/pass  graph {msg*1e3:8.1f} us/pass  -> {el/msg/1e3:8.1f} M element-solves/s "
              f"(elements {el}, with root {(nr>0).sum()}, brackets>=1.. roots total {int(nr.sum())})", flush=True)

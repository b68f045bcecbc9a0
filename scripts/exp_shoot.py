import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rtus
from importlib import import_module
dev_api=import_module("ray-tracing-ultrasound_amd.device")
dev=torch.device('cuda')
t64=lambda a: torch.as_tensor(np.ascontiguousarray(a,dtype=np.float64),device=dev)
d=rtus.Params().d
for N, fast in [(n, f) for n in (905, 8192, 32768) for f in (False, True)]:
    T=max(1,(8*1024*1024)//N)
    xa=(np.arange(T)-(T-1)/2)*(0.04/T)
    alpha=np.linspace(-rtus.ALPHA_MAX,rtus.ALPHA_MAX,N)
    plan=dev_api.ShootPlan(1,T,N,want=("tof","land_x"),params=rtus.Params(),fast=fast)
    a=[t64([[0.037,0.0038]]),t64(xa),t64(np.full(T,d)),t64(alpha),t64(np.full(N,d))]
    for _ in range(3): plan.run(*a)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): plan.run(*a)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    lx=plan.out["land_x"]
    print(f"fast={int(fast)} N={N:6d} T={T:6d} rays={T*N:9d}  {ms*1e3:9.1f} us  {T*N/ms/1e6:8.2f} Grays/s  finite={float(torch.isfinite(lx).double().mean()):.3f}")

import os, sys, ctypes, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rtus
from importlib import import_module
dev_api=import_module("ray-tracing-ultrasound_amd.device")
L=rtus.lib()
dev=torch.device('cuda')
t64=lambda a: torch.as_tensor(np.ascontiguousarray(a,dtype=np.float64),device=dev)
d=rtus.Params().d
sym=ctypes.c_void_p(); sz=ctypes.c_size_t()
for N in (905, 8192, 32768):
    T=max(1,(1024*1024)//N)
    xa=(np.arange(T)-(T-1)/2)*(0.04/T)
    alpha=np.linspace(-rtus.ALPHA_MAX,rtus.ALPHA_MAX,N)
    plan=dev_api.ShootPlan(1,T,N,want=("tof","land_x"),params=rtus.Params(),fast=True)
    a=[t64([[0.037,0.0038]]),t64(xa),t64(np.full(T,d)),t64(alpha),t64(np.full(N,d))]
    buf=(ctypes.c_ulonglong*8)()
    # zero the counters via hipMemcpyToSymbol-equivalent: get symbol address from the module
    L.rtus_dbg_read(buf, 1)
    plan.run(*a); torch.cuda.synchronize()
    L.rtus_dbg_read(buf, 1)
    waves=T*((N+255)//256)*4
    print(f"N={N} waves={waves}", "per wave: L2certs %.1f L1certs %.1f L0certs %.1f leaves %.1f | pass2 frac %.2f L2 %.1f L0 %.1f leaves %.1f"%tuple(np.array(list(buf))/waves))

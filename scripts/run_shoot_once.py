import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rtus
from importlib import import_module
dev_api=import_module("ray-tracing-ultrasound_amd.device")
dev=torch.device('cuda')
t64=lambda a: torch.as_tensor(np.ascontiguousarray(a,dtype=np.float64),device=dev)
d=rtus.Params().d
fast=int(sys.argv[1]) if len(sys.argv)>1 else 1
N=8192; T=1024
xa=(np.arange(T)-(T-1)/2)*(0.04/T)
alpha=np.linspace(-rtus.ALPHA_MAX,rtus.ALPHA_MAX,N)
plan=dev_api.ShootPlan(1,T,N,want=("tof","land_x"),params=rtus.Params(),fast=bool(fast))
a=[t64([[0.037,0.0038]]),t64(xa),t64(np.full(T,d)),t64(alpha),t64(np.full(N,d))]
for _ in range(3): plan.run(*a)
torch.cuda.synchronize()

import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rtus
from importlib import import_module
dev_api=import_module("ray-tracing-ultrasound_amd.device")
dev=torch.device('cuda')
t64=lambda a: torch.as_tensor(np.ascontiguousarray(a,dtype=np.float64),device=dev)
def run(n_e,g,z_if,c,iters=False,reps=50):
    xe=(np.arange(n_e)-(n_e-1)/2)*0.6e-3*128/n_e
    xs,zs=np.meshgrid(np.linspace(-0.02,0.02,g),np.linspace(0.026,0.065,g))
    a=[t64(xe),t64(np.zeros(n_e)),t64(xs.ravel()),t64(zs.ravel())]
    out=torch.empty((n_e,g*g),dtype=torch.float64,device=dev)
    it=torch.empty((n_e,g*g),dtype=torch.uint8,device=dev) if iters else None
    for _ in range(5): dev_api.tt_layers_dev(z_if,c,*a,out=out,iters=it)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): dev_api.tt_layers_dev(z_if,c,*a,out=out,iters=it)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/reps
    h=np.bincount(it.cpu().numpy().ravel(),minlength=8)[:8] if iters else None
    return ms,h
for n_e in (8,32,128,512,2048):
    ms,h=run(n_e,128,[0.02],[2330.,1483.],iters=True)
    ms2,_=run(n_e,128,[0.02],[2330.,1483.])
    print(f"n_e={n_e:5d} g=128 NL=2  {ms2*1e3:8.2f} us  ({n_e*16384/ms2/1e6:9.1f} Gsolves/s... ) with iters out {ms*1e3:8.2f} us  iter hist {h}")
ms,h=run(256,512,[0.01,0.025],[2330.,1483.,5900.],iters=True,reps=5); print('cfg3',ms,h)

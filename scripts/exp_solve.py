import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtus
d=rtus.Params().d
for T,E,N in ((65,65,905),(256,256,905),(1024,1024,905),(1024,1024,4096)):
    x=(np.arange(T)-(T-1)/2)*(0.0384/T); xr=(np.arange(E)-(E-1)/2)*(0.0384/E)
    alpha=np.linspace(-rtus.ALPHA_MAX,rtus.ALPHA_MAX,N)
    p=rtus.Params(r_outer=0.05,pipe_offset=0.0)
    for fast in (False,True):
        rtus.solve_travel_times(x,np.full(T,d),xr,alpha,params=p,fast=fast)
        t0=time.perf_counter(); tt,ar=rtus.solve_travel_times(x,np.full(T,d),xr,alpha,params=p,fast=fast); dt=time.perf_counter()-t0
        print(f"T={T} E={E} N={N} fast={int(fast)}: {dt*1e3:8.2f} ms host API, roots {np.isfinite(tt).sum()} of {tt.size}  -> {tt.size/dt/1e6:.2f} M elements/s")

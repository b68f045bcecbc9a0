import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtus
d=rtus.Params().d
T=E=1024; N=int(sys.argv[1]); fast=bool(int(sys.argv[2]))
x=(np.arange(T)-(T-1)/2)*(0.0384/T)
alpha=np.linspace(-rtus.ALPHA_MAX,rtus.ALPHA_MAX,N)
p=rtus.Params(r_outer=0.05,pipe_offset=0.0)
for _ in range(2):
    tt,ar,ta,aa,nr=rtus.solve_travel_times(x,np.full(T,d),x,alpha,params=p,fast=fast,all_roots=True)
print(N,fast,np.bincount(nr.ravel(),minlength=5))

import os, sys, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtus
L=rtus.lib()
d=rtus.Params().d
buf=(ctypes.c_ulonglong*8)()
T=256; N=905
xa=(np.arange(T)-(T-1)/2)*(0.0384/T)
alpha=np.linspace(-rtus.ALPHA_MAX,rtus.ALPHA_MAX,N)
for off in (0.0, 0.0038):
    for fast in (False, True):
        L.rtus_dbg_read(buf,1)
        b=rtus.shoot_batch(xa,np.full(T,d),np.full(N,d),alpha,params=rtus.Params(r_outer=0.05,pipe_offset=off),want=("land_x",),fast=fast)
        L.rtus_dbg_read(buf,1)
        waves=T*((N+255)//256)*4
        print(f"off={off} fast={int(fast)} waves={waves}", "per wave: L2 %.1f L1 %.1f L0 %.1f gathers(passes) %.2f | pass2 frac %.2f L2 %.1f L0 %.1f leaves %.1f"%tuple(np.array(list(buf))/waves), 'finite', np.isfinite(b['land_x']).mean())

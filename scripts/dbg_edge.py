import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtus
e=np.load('tests/golden/edge_cfg.npz')
D=float(np.float64(0.12156646438729327)+np.float64(0.08843353561270673))
for tag in ("q1nan","tir","off0","offtx"):
    r_o,off,x_tx=e[tag+'_cfg']
    p=rtus.Params(r_outer=float(r_o),pipe_offset=float(off))
    res=rtus.shoot_rays(float(x_tx),D,np.full(905,D),e['alpha'],params=p)
    o=np.stack([res[k] for k in rtus.KEYS]); r=e[tag]
    for k in range(8):
        dm=np.where(np.isnan(o[k])!=np.isnan(r[k]))[0]
        print(tag,k,'nanmask diff at',dm[:20],len(dm))
        if len(dm): print('  gpu',o[k][dm[:5]],'ref',r[k][dm[:5]])

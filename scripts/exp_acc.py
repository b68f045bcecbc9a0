import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rtus
from oracle import cport
xe=(np.arange(128)-63.5)*0.6e-3; ze=np.zeros(128)
xs,zs=np.meshgrid(np.linspace(-0.02,0.02,128),np.linspace(0.025,0.065,128)); xf,zf=xs.ravel(),zs.ravel()
for c in ((2330.,1483.),(1483.,5900.)):
    tt=rtus.travel_time_layers([0.02],c,xe,ze,xf,zf); ref=cport.tt_layers([0.02],c,xe[::4],ze[::4],xf,zf)
    e=np.abs(tt[::4]-ref); print(c,'max abs err',e.max(),'rel',(e/ref).max(),'median',np.median(e))
z_if,c=[0.010,0.025],[2330.,1483.,5900.]
xe=(np.arange(64)-31.5)*0.3e-3
xs,zs=np.meshgrid(np.linspace(-0.03,0.03,96),np.linspace(0.001,0.06,80))
tt=rtus.travel_time_layers(z_if,c,xe,np.zeros(64),xs.ravel(),zs.ravel()); ref=cport.tt_layers(z_if,c,xe,np.zeros(64),xs.ravel(),zs.ravel())
e=np.abs(tt-ref); print('cfg3-like max abs err',e.max(),'rel',(e/ref).max())

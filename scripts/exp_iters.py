"""Planar kernel: Newton steps per position of the element inside its workgroup block (cfg2 shape)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtus
xe = (np.arange(128) - 63.5) * 0.6e-3
xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 128), np.linspace(0.026, 0.065, 128))
for name, z_if, c in (("cfg2", [0.02], [2330., 1483.]), ("slow-over-fast", [0.02], [1483., 5900.]),
                      ("3 layers", [0.01, 0.025], [2330., 1483., 5900.])):
    tt, it = rtus.travel_time_layers(z_if, c, xe, np.zeros(128), xs.ravel(), zs.ravel(), return_iters=True)
    it = it.reshape(128, -1, 64)                       # element, wave, lane
    print(name, "mean steps/lane by element 0..15:", np.round(it.mean(axis=(1, 2))[:16], 2))
    print(name, "mean wave-max steps by element 0..15:", np.round(it.max(axis=2).mean(axis=1)[:16], 2),
          " overall wave-max mean", round(float(it.max(axis=2).mean()), 3))

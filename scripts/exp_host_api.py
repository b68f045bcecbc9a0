"""Host-buffer API (NumPy in, NumPy out — the drop-in boundary) on BASELINE config 2: PCIe-inclusive time per call,
fresh result array vs a caller-owned one.  (A page-locked result buffer was tried and made no difference on this
box: 0.614 vs 0.618 ms — the pageable copy already runs at the link rate, 27 GB/s.)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtus
xe = (np.arange(128) - 63.5) * 0.6e-3
xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 128), np.linspace(0.025, 0.065, 128))
args = ([0.02], [2330.0, 1483.0], xe, np.zeros(128), xs.ravel(), zs.ravel())
n = 128 * 16384
def best(fn, k=20):
    fn(); fn()
    b = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return b
pg = np.empty((128, 16384))
for name, fn in (("fresh pageable result (default)", lambda: rtus.travel_time_layers(*args)),
                 ("caller's out=", lambda: rtus.travel_time_layers(*args, out=pg))):
    t = best(fn)
    print(f"{name:34s} {t * 1e3:8.3f} ms/call  {n / t / 1e6:9.1f} Mrays/s  {n * 8 / t / 1e9:6.2f} GB/s of results")

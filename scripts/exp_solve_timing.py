"""Where the solve kernel's time goes: an RTUS_EXP_TIMING build (scripts/build_variant.sh timing -DRTUS_EXP_TIMING) leaves
100 MHz wall-clock stamps per wave at the phase boundaries; one sweep-sized pass, then the table."""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTUS_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants", "librtus_timing.so"))
import numpy as np, torch, rtus
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
d = rtus.Params().d
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
n = 905
alpha = t64(np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)); x_rx = t64(rtus.reference_elements())
geoms = np.array([[r * 1e-2, o * 1e-3] for r in range(1, 11) for o in range(-10, 11)]); xa = np.array([0.0])
plan = dev_api.SolvePlan(len(geoms), 1, n, 65, params=rtus.Params())
a = (t64(geoms), t64(xa), t64(np.full(1, d)), alpha, x_rx)
for _ in range(5):
    plan.run(*a)
torch.cuda.synchronize()
buf = np.zeros((4096, 16), dtype=np.uint64)
L = rtus.lib()
L.rtus_solve_stamps_read.argtypes = [C.c_void_p]
assert L.rtus_solve_stamps_read(buf.ctypes.data) == 0
nw = int((buf[:, 0] > 0).sum())
s = buf[:nw].astype(np.int64)
t0 = s[:, 0].min()
names = ["entry", "A done (row kernel: trace done)", "B done"] + [f"eval {i}" for i in range(10)] + ["C done", "barrier", "end"]
print("wave-level stamps, microseconds after the first wave's entry (100 MHz clock); waves that skip a phase keep 0")
for i, nm in enumerate(names):
    v = s[:, i]; ok = v > 0
    if ok.any():
        u = (v[ok] - t0) / 100.0
        print(f"  {nm:8s} waves {ok.sum():4d}  min {u.min():7.2f}  median {np.median(u):7.2f}  max {u.max():7.2f}")
# per-wave eval durations
ev = s[:, 3:13]
dur = []
for w in range(nw):
    st = [x for x in ev[w] if x > 0]
    st.append(s[w, 13])
    dur += [(st[i + 1] - st[i]) / 100.0 for i in range(len(st) - 1) if st[i + 1] > st[i]]
dur = np.array(dur)
print(f"  evaluation duration per wave: n {dur.size} min {dur.min():.2f} median {np.median(dur):.2f} mean {dur.mean():.2f} max {dur.max():.2f} us")
nev = (ev > 0).sum(1)
print("  evaluations per wave:", np.bincount(nev))

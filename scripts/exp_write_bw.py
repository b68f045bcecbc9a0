"""Device write bandwidth seen by simple store kernels (torch fill / copy), to compare with the 2.9-3.0 TB/s of
results the planar kernel sustains on large launches."""
import sys, os, torch
dev = torch.device("cuda")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mb in (16.8, 67, 537, 2148):
    n = int(mb * 1e6 / 8)
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(30): a.fill_(1.0)          # clock warm-up
    ms_fill = timeit(lambda: a.fill_(1.5))
    ms_copy = timeit(lambda: b.copy_(a))
    ms_add = timeit(lambda: torch.add(a, 1.0, out=b))
    print(f"{mb:7.1f} MB: fill {ms_fill * 1e3:8.1f} us = {n * 8 / ms_fill / 1e9:6.2f} TB/s written | copy {ms_copy * 1e3:8.1f} us = "
          f"{n * 8 / ms_copy / 1e9:6.2f} TB/s written (+ same read) | add {ms_add * 1e3:8.1f} us = {n * 8 / ms_add / 1e9:6.2f} TB/s written", flush=True)

"""A/B of librtus.so builds on the curved-lens Fermat kernels (BASELINE configs[3] shard: 128 elements x 1024 x 1024
targets, fp32; 1024 x (1024 x 256) fp64), interleaved rounds in one process: python scripts/ab_lens.py libA.so libB.so"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rtus  # noqa: E402


def load(path):
    L = C.CDLL(os.path.abspath(path), mode=C.RTLD_LOCAL)
    dp, ip, vp = C.c_void_p, C.c_int, C.c_void_p
    for n in ("rtus_tt_lens_dev", "rtus_tt_lens_f32_dev"):
        getattr(L, n).argtypes = [C.POINTER(rtus.Lens), C.c_double, C.c_double, dp, dp, ip, dp, dp, ip, dp, dp, vp]
        getattr(L, n).restype = ip
    return L


def main():
    paths = [p for p in sys.argv[1:] if p.endswith(".so")]
    libs = [load(p) for p in paths]
    dev = torch.device("cuda", 0)
    lens = rtus.Params().lens()
    for name, rows, nf, dt, fn in (("f32 128 x 1M", 128, 1 << 20, np.float32, "rtus_tt_lens_f32_dev"),
                                   ("f32 1024 x 1M", 1024, 1 << 20, np.float32, "rtus_tt_lens_f32_dev"),
                                   ("f64 256 x 256K", 256, 1 << 18, np.float64, "rtus_tt_lens_dev")):
        W = bench.lens_inputs(0, 1, n_rows=rows)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=dev)
        xe, ze, xf, zf = t(W["xe"]), t(W["ze"]), t(W["xf"][:nf]), t(W["zf"][:nf])
        outs = [torch.empty((rows, nf), dtype=xe.dtype, device=dev) for _ in libs]

        def run(i):
            st = getattr(libs[i], fn)(C.byref(lens), -rtus.ALPHA_MAX, rtus.ALPHA_MAX, xe.data_ptr(), ze.data_ptr(), rows, xf.data_ptr(),
                                      zf.data_ptr(), nf, outs[i].data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            assert st == 0, st

        for i in range(len(libs)):
            for _ in range(2):
                run(i)
        torch.cuda.synchronize()
        times = [[] for _ in libs]
        reps = 3 if rows >= 1024 else 10
        for r in range(7):
            for i in range(len(libs)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(i)
                e1.record(); torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / reps)
        line = f"{name}: "
        for i, p in enumerate(paths):
            m = float(np.median(times[i]))
            d = float(torch.max(torch.abs(outs[i].double() - outs[0].double())).item())
            line += f"[{os.path.basename(p)}] {m*1e3:9.1f} us  {rows*nf/m/1e3:9.0f} Mrays/s  max|dt| vs A {d:.1e}   "
        print(line, flush=True)


if __name__ == "__main__":
    main()

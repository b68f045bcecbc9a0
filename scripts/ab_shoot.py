"""A/B of librtus.so builds on the reference-geometry forward trace (1024 tx x 8192 rays, tof + land_x), interleaved
rounds in one process: python scripts/ab_shoot.py libA.so libB.so [...]"""
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
    L.rtus_shoot_dev.argtypes = [C.POINTER(rtus.Lens), dp, ip, dp, dp, ip, dp, dp, ip, dp, dp, dp, dp, vp, vp, C.c_size_t, C.c_uint, vp]
    L.rtus_shoot_dev.restype = ip
    L.rtus_shoot_workspace_bytes.argtypes = [ip]
    L.rtus_shoot_workspace_bytes.restype = C.c_size_t
    return L


def main():
    paths = [p for p in sys.argv[1:] if p.endswith(".so")]
    libs = [load(p) for p in paths]
    dev = torch.device("cuda", 0)
    t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    for kind, flags in (("ref_scale", 0), ("ref_scale", 1), ("ref_sweep", 0)):
        R = bench.ref_inputs(kind)
        G, T, N = R["geoms"].shape[0], R["xa"].size, R["n"]
        geoms, xa, za, alpha, zf = t64(R["geoms"]), t64(R["xa"]), t64(R["za"]), t64(R["alpha"]), t64(R["zf"])
        lens = rtus.Params().lens()
        outs = []
        for L in libs:
            wsb = L.rtus_shoot_workspace_bytes(N)
            outs.append((torch.empty(wsb, dtype=torch.uint8, device=dev), torch.empty((G, T, N), dtype=torch.float64, device=dev),
                         torch.empty((G, T, N), dtype=torch.float64, device=dev), wsb))

        def run(i):
            ws, tof, lx, wsb = outs[i]
            st = libs[i].rtus_shoot_dev(C.byref(lens), geoms.data_ptr(), G, xa.data_ptr(), za.data_ptr(), T, alpha.data_ptr(),
                                        zf.data_ptr(), N, None, None, tof.data_ptr(), lx.data_ptr(), None, ws.data_ptr(), wsb, flags,
                                        torch.cuda.current_stream().cuda_stream)
            assert st == 0, st

        times = [[] for _ in libs]
        for i in range(len(libs)):
            for _ in range(3):
                run(i)
        torch.cuda.synchronize()
        reps = 5 if kind == "ref_scale" else 50
        for r in range(7):
            for i in range(len(libs)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(i)
                e1.record(); torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / reps)
        base = outs[0]
        line = f"{kind} flags={flags}: "
        for i, p in enumerate(paths):
            m = float(np.median(times[i]))
            d = torch.nan_to_num(torch.abs(outs[i][2] - base[2]), nan=0.0).max().item()
            nm = int((torch.isnan(outs[i][2]) != torch.isnan(base[2])).sum().item())
            line += f"[{os.path.basename(p)}] {m*1e3:8.1f} us  {G*T*N/m/1e6:7.2f} G rays/s  max|dx_land| vs A {d:.1e} nan-mismatch {nm}   "
        print(line, flush=True)


if __name__ == "__main__":
    main()

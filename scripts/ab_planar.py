"""A/B of two builds of librtus.so on the planar Fermat workloads, interleaved rounds in ONE process
(cdna_hip_programming.md rule 24): python scripts/ab_planar.py scripts/librtus_r01.so ray-tracing-ultrasound_amd/librtus.so
Prints per workload: median / min ms per launch for each build, the ratio, and the max |dt| between the two results and
against the long-double oracle on a seeded sample."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def load(path):
    L = C.CDLL(os.path.abspath(path), mode=C.RTLD_LOCAL)
    dp, ip, vp = C.c_void_p, C.c_int, C.c_void_p
    L.rtus_tt_layers_dev.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, vp, vp]
    L.rtus_tt_layers_dev.restype = ip
    L.rtus_tt_layers_ex_dev.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, vp, C.c_uint, vp]
    L.rtus_tt_layers_ex_dev.restype = ip
    return L


TIER = 1 if os.environ.get("AB_TAUP") else 0          # RTUS_TT_TAUP_TAIL: the tier the headline times


def main():
    paths = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    libs = [load(p) for p in paths]
    dev = torch.device("cuda", 0)
    t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    from oracle import cport
    for wl, reps in (("cfg3_planar", 20), ("cfg2_planar", 200), ("cfg5_fmc", 100)):
        W = bench.workload_inputs(wl, 0, 1)
        z_if = np.ascontiguousarray(W["z_if"], dtype=np.float64); c = np.ascontiguousarray(W["c"], dtype=np.float64)
        xe, ze, xf, zf = t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"])
        outs = [torch.empty((W["n_e"], W["n_f"]), dtype=torch.float64, device=dev) for _ in libs]
        def run(i):
            st = libs[i].rtus_tt_layers_ex_dev(z_if.ctypes.data, c.ctypes.data, z_if.size, xe.data_ptr(), ze.data_ptr(), W["n_e"],
                                               xf.data_ptr(), zf.data_ptr(), W["n_f"], outs[i].data_ptr(), None, TIER,
                                               torch.cuda.current_stream().cuda_stream)
            assert st == 0, st

        graphs = []
        for i in range(len(libs)):
            for _ in range(3):
                run(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    run(i)
            g.replay(); torch.cuda.synchronize()
            graphs.append(g)
        times = [[] for _ in libs]
        for r in range(rounds):
            for i in range(len(libs)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); graphs[i].replay(); e1.record(); torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / reps)
        med = [float(np.median(t)) for t in times]
        mn = [float(np.min(t)) for t in times]
        n = W["n_e"] * W["n_f"]
        rng = np.random.default_rng(7)
        re = np.sort(rng.choice(W["n_e"], size=min(16, W["n_e"]), replace=False))
        cf = np.sort(rng.choice(W["n_f"], size=min(4096, W["n_f"]), replace=False))
        ref = cport.tt_layers(W["z_if"], W["c"], W["xe"][re], W["ze"][re], W["xf"][cf], W["zf"][cf])
        errs = []
        for o in outs:
            got = o[torch.as_tensor(re, device=dev)][:, torch.as_tensor(cf, device=dev)].cpu().numpy()
            errs.append(float(np.nanmax(np.abs(got - ref))))
        dab = float(torch.max(torch.abs(outs[0] - outs[1])).item()) if len(outs) == 2 else 0.0
        print(f"{wl}: " + "  ".join(f"[{os.path.basename(p)}] median {m*1e3:8.2f} us  min {k*1e3:8.2f} us  {n/m/1e3:9.0f} Mrays/s  "
                                    f"hbm {n*8/(m*1e-3)/8e12:.4f}  max|dt| vs oracle {e:.2e}"
                                    for p, m, k, e in zip(paths, med, mn, errs)) +
              (f"  | B/A time {med[1]/med[0]:.3f}  max|A-B| {dab:.2e} s" if len(libs) == 2 else ""), flush=True)


if __name__ == "__main__":
    main()

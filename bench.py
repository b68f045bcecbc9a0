#!/usr/bin/env python3
"""bench.py — Mrays/s (element x focal-point travel-time solves per second) on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM.  The default
workload is BASELINE.json configs[1] — 128-element linear array, 1 planar interface, 128x128 focal grid,
fp64 — i.e. 2,097,152 Fermat travel-time solves per step per GPU (rtus_tt_layers_dev,
csrc/rtus_fermat.hip).  The K timed steps are K back-to-back launches on one stream, captured once into a
hipGraph and replayed (--graph off: eager launches).

--gpus N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): every rank solves its own
128-element block of a 128*N-element aperture (weak scaling; the solves need no exchange, so there is no
collective on the timed path).  The [128*N, 16384] travel-time matrix is then reassembled on every rank by ONE
RCCL all-gather, timed on its own and reported as `reassembly` (--gather after, default).  --gather end puts
that all-gather inside the timed region, --gather step gathers after every step on a double-buffered matrix
overlapped with the next kernel, --gather off never gathers.  (An all-gather per step cannot keep up with the
kernel: a GPU produces ~1.3 TB/s of results and would have to receive 7x that.)

Other workloads (--workload): cfg3_planar (configs[2]), cfg4_lens_f32 (configs[3], curved lens, fp32),
cfg5_fmc (configs[4], FMC table), ref_sweep (the reference's own sweep, main_rt.py:464-501: 210 geometries x
905 rays forward trace + 65-element matcher), ref_scale (reference geometry, 1024 tx x 8192 rays).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against the HBM roof as the contract
asks (8 B written per solve; `traffic` = PMC-measured HBM bytes per launch from profiles/traffic_r01.json)
and says which bound actually binds (VALU issue).  `cpu_baseline` is the oracle's fp64 CPU port timed on
this box's host cores (rank 0, N=1 only); `extra` holds side measurements: the reference-parity path, the same
planar kernel on configs[2] and the curved-lens kernel of configs[3].
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREWARM_MS = 30.0              # untimed graph replays before the timed region (clock ramp), disclosed in config
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)   # 2000 x ~10 us: a 20 ms timed region (barrier / graph-launch latency < 1 %)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg2_planar",
                    choices=["cfg2_planar", "cfg3_planar", "cfg4_lens_f32", "cfg5_fmc", "ref_sweep", "ref_scale"])
    ap.add_argument("--gather", default="after", choices=["after", "end", "step", "off"],
                    help="--gpus > 1: RCCL all-gather of the row shards — once AFTER the timed region, timed on its own "
                         "(after, default: the solves need no exchange, the reassembly is reported separately); once "
                         "inside the timed region after the K steps (end); after every step, overlapped with the next "
                         "kernel (step); never (off)")
    ap.add_argument("--graph", default="on", choices=["on", "off"],
                    help="replay the K timed steps as one captured hipGraph (N=1 or --gather end/off)")
    ap.add_argument("--streams", type=int, default=1,
                    help="planar workloads, graph mode: capture the K steps round-robin on this many streams, each with its "
                         "own output buffer, so that one launch's ramp-up overlaps the previous one's drain (default 1: "
                         "K back-to-back launches on one stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the reference-path side measurements")
    return ap.parse_args()


# ------------------------------------------------------------------------------------ workloads
def planar_inputs(cfg, rank, world):
    """cfg2 = 128 elems @0.6 mm on z=0, interface z=20 mm, c=(2330,1483), 128x128
    focal grid over x in [-20,20] mm, z in [25,65] mm; cfg3 = 256 elems @0.3 mm, interfaces at
    10/25 mm, c=(2330,1483,5900), 512x512 grid.  Rank r owns elements [r*n_e, (r+1)*n_e) of a
    world*n_e-element aperture (weak scaling)."""
    if cfg == "cfg2_planar":
        n_e, pitch, z_if, c, g, zr = 128, 0.6e-3, [0.020], [2330.0, 1483.0], 128, (0.025, 0.065)
    else:
        n_e, pitch, z_if, c, g, zr = 256, 0.3e-3, [0.010, 0.025], [2330.0, 1483.0, 5900.0], 512, (0.026, 0.066)
    n_all = n_e * world
    x_all = (np.arange(n_all) - (n_all - 1) / 2.0) * pitch
    xe = x_all[rank * n_e:(rank + 1) * n_e]
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, g), np.linspace(zr[0], zr[1], g))
    return dict(z_if=z_if, c=c, xe=xe, ze=np.zeros(n_e), xf=xs.ravel(), zf=zs.ravel(), n_e=n_e, n_f=g * g)


def lens_inputs(rank, world):
    """BASELINE configs[3]: 1024-element array @0.3 mm on z = d over the reference lens h(alpha),
    water below, 1024 x 1024 target grid inside the insonified cone, fp32; tx rows sharded 1024/world per GPU
    (this config is a STRONG-scaling shape in BASELINE; here each rank keeps 128 rows = the 8-GPU shard)."""
    import rtus
    n_e = 128
    x_all = (np.arange(n_e * world) - (n_e * world - 1) / 2.0) * 0.3e-4
    xs, zs = np.meshgrid(np.linspace(-0.004, 0.004, 1024), np.linspace(0.03, 0.07, 1024))
    return dict(xe=x_all[rank * n_e:(rank + 1) * n_e], ze=np.full(n_e, rtus.Params().d), xf=xs.ravel(), zf=zs.ravel(),
                n_e=n_e, n_f=xs.size)


def fmc_inputs(rank, world):
    """BASELINE configs[4]: 2048 tx x 2048 rx on z = 0, 3 horizontal layers, planar reflector at
    depth — unfolded about the reflector into a 5-layer one-way problem; each rank owns 256 tx rows."""
    z_if, c, z_r = np.array([0.008, 0.020]), np.array([2330.0, 1483.0, 5900.0]), 0.035
    z_m = np.concatenate([z_if, (2.0 * z_r - z_if)[::-1]])
    c_m = np.concatenate([c, c[::-1][1:]])
    n_e = 256
    x_tx = (np.arange(n_e * world) - (n_e * world - 1) / 2.0) * 0.3e-3
    x_rx = (np.arange(2048) - 1023.5) * 0.3e-3
    return dict(z_if=z_m, c=c_m, xe=x_tx[rank * n_e:(rank + 1) * n_e], ze=np.zeros(n_e), xf=x_rx,
                zf=np.full(2048, 2.0 * z_r), n_e=n_e, n_f=2048)


def ref_inputs(kind):
    import rtus
    d = rtus.Params().d
    if kind == "ref_sweep":          # main_rt.py:464-482
        n = 905
        geoms = np.array([[r * 1e-2, o * 1e-3] for r in range(1, 11) for o in range(-10, 11)])
        xa = np.array([0.0])
    else:                            # reference geometry scaled up (SURVEY 8(d) row "R")
        n = 8192
        geoms = np.array([[0.037, 0.0038]])
        xa = (np.arange(1024) - 511.5) * 0.3e-3
    alpha = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)
    return dict(n=n, geoms=geoms, xa=xa, za=np.full(xa.size, d), alpha=alpha, zf=np.full(n, d),
                x_rx=rtus.reference_elements())


# ------------------------------------------------------------------------------------ main
def main():
    args = parse()
    # the CPU-baseline leg runs an OpenMP port on every host core; idle OpenMP workers must sleep, not spin, while the
    # GPU side measurements that follow are launched from this thread
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    import rtus
    from importlib import import_module
    dev_api = import_module("ray-tracing-ultrasound_amd.device")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    # Rehearsal switches (one-GPU box): RTUS_BENCH_ONE_GPU=1 puts every rank on cuda:0 and
    # RTUS_BENCH_BACKEND=gloo swaps RCCL for gloo, so the N>1 control flow can be exercised without N GPUs.
    one_gpu = os.environ.get("RTUS_BENCH_ONE_GPU", "0") == "1"
    backend = os.environ.get("RTUS_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    wl = args.workload
    gather = world > 1 and args.gather == "step"
    gather_end = world > 1 and args.gather == "end"
    gather_after = world > 1 and args.gather == "after"

    if wl == "cfg4_lens_f32":
        import ctypes as C
        W = lens_inputs(rank, world)
        t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
        xe, ze, xf, zf = t32(W["xe"]), t32(W["ze"]), t32(W["xf"]), t32(W["zf"])
        n_e, n_f = W["n_e"], W["n_f"]
        dist_api = import_module("ray-tracing-ultrasound_amd.dist")
        slots = 2 if gather else 1
        m = dist_api.RowShardedMatrix(world * n_e, n_f, dtype=torch.float32, device=dev, slots=slots)
        lens = rtus.Params().lens()
        fn = rtus.lib().rtus_tt_lens_f32_dev
        units_per_step = n_e * n_f
        alg_bytes = units_per_step * 4 + (2 * n_e + 2 * n_f) * 4
        kernel = "rtus_tt_lens_kernel<float>"

        def step(s):
            b = s % slots
            if gather:
                m.wait(b)
            st = fn(C.byref(lens), -rtus.ALPHA_MAX, rtus.ALPHA_MAX, xe.data_ptr(), ze.data_ptr(), n_e, xf.data_ptr(),
                    zf.data_ptr(), n_f, m.local(b).data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            assert st == 0
            if gather:
                m.gather(b, async_op=True)

        def drain():
            for b in range(slots):
                m.wait(b)

        def finish(force=False):
            if gather_end or force:
                m.gather(0)
    elif wl in ("cfg2_planar", "cfg3_planar", "cfg5_fmc"):
        W = fmc_inputs(rank, world) if wl == "cfg5_fmc" else planar_inputs(wl, rank, world)
        xe, ze, xf, zf = t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"])
        n_e, n_f = W["n_e"], W["n_f"]
        # double-buffered full matrix; each rank's kernel writes straight into its row block
        dist_api = import_module("ray-tracing-ultrasound_amd.dist")
        slots = 2 if gather else max(1, args.streams)
        m = dist_api.RowShardedMatrix(world * n_e, n_f, device=dev, slots=slots)
        plans = [dev_api.LayersPlan(W["z_if"], W["c"], xe, ze, xf, zf, out=m.local(b)) for b in range(slots)]
        units_per_step = n_e * n_f
        alg_bytes = units_per_step * 8 + (2 * n_e + 2 * n_f) * 8
        kernel = f"rtus_tt_layers_kernel<{len(W['c'])}, false>"

        def step(s):
            b = s % slots
            if gather:
                m.wait(b)               # the slot's previous all-gather must finish before it is rewritten
            plans[b].run()
            if gather:
                m.gather(b, async_op=True)      # RCCL, overlaps the next step's kernel

        def drain():
            for b in range(slots):
                m.wait(b)

        def finish(force=False):        # reassemble the last step's matrix on every rank (RCCL all-gather over xGMI)
            if gather_end or force:
                m.gather(0)
    else:
        R = ref_inputs(wl)
        G, T, N = R["geoms"].shape[0], R["xa"].size, R["n"]
        plan = dev_api.ShootPlan(G, T, N, want=("tof", "land_x"), params=rtus.Params(), device=dev)
        geoms, xa, za, alpha, zf = t64(R["geoms"]), t64(R["xa"]), t64(R["za"]), t64(R["alpha"]), t64(R["zf"])
        x_rx = t64(R["x_rx"])
        mout = None
        units_per_step = G * T * N
        alg_bytes = units_per_step * 16 + N * 16       # tof + land_x written per ray; alpha, z_f read once
        kernel = "rtus_shoot_kernel"

        def step(s):
            nonlocal mout
            o = plan.run(geoms, xa, za, alpha, zf)
            mout = dev_api.match_dev(o["land_x"].view(G * T, N), o["tof"].view(G * T, N), x_rx, 1e-6, 1e-5, out=mout)

        def drain():
            pass

        def finish(force=False):
            pass

    for s in range(args.warmup):
        step(s)
    gather_error = None
    try:
        finish(force=gather_after)      # also warms the communicator up
    except Exception as exc:            # the timed region needs no collective: keep measuring, report the failure
        if not gather_after:
            raise
        gather_error = repr(exc)
        print(f"[bench] warm-up all-gather failed: {gather_error}", file=sys.stderr)
    drain()
    torch.cuda.synchronize()

    # The K timed steps are K back-to-back launches on one stream.  Python + hipLaunchKernel cost
    # ~10 us per call, comparable to the kernel itself on cfg2, so the K launches are captured once
    # into a hipGraph and replayed (same kernels, same order, no per-launch host work).
    graph = None
    if args.graph == "on" and not gather:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                # thread_local: other threads (e.g. the RCCL watchdog at --gpus > 1) may keep issuing HIP calls
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    if args.streams > 1 and wl in ("cfg2_planar", "cfg3_planar", "cfg5_fmc"):
                        subs = [torch.cuda.Stream() for _ in range(args.streams)]     # fork ...
                        for st in subs:
                            st.wait_stream(side)
                        for s in range(args.steps):
                            with torch.cuda.stream(subs[s % args.streams]):             # step s -> stream / buffer s % S
                                step(s)
                        for st in subs:                                                   # ... and join
                            side.wait_stream(st)
                    else:
                        for s in range(args.steps):
                            step(s)
            torch.cuda.current_stream().wait_stream(side)
            # untimed replays: the first uploads the graph; then keep the GPU busy for PREWARM_MS so that the timed
            # region starts at the sustained clock whatever K is (the clocks of an idle MI355X take a few ms of load
            # to come up: a 200-step region measured cold reads ~10 % slower than a 2000-step one)
            t_pre = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            one = max(time.perf_counter() - t_pre, 1e-5)
            for _ in range(min(200, int(PREWARM_MS * 1e-3 / one))):
                g.replay()
            torch.cuda.synchronize()
            graph = g
        except Exception as exc:            # capture unsupported -> eager launches
            print(f"[bench] hipGraph capture failed ({exc!r}); timing eager launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()                      # same (current) stream the library launches on
    if graph is not None:
        graph.replay()
    else:
        for s in range(args.steps):
            step(s)
    ev1.record()
    finish()
    drain()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # average launch duration over the timed region
    reassembly_ms = None
    if gather_after and gather_error is None:         # the one exchange of the path, outside the solve loop
        try:
            t1 = time.perf_counter()
            finish(force=True)
            torch.cuda.synchronize()
            barrier()
            reassembly_ms = (time.perf_counter() - t1) * 1e3
        except Exception as exc:
            gather_error = repr(exc)
            print(f"[bench] all-gather failed: {gather_error}", file=sys.stderr)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_units = units_per_step * world * args.steps
    value = total_units / dt / 1e6

    out = {
        "metric": "Mrays/sec (elem x focal travel-time solves)", "value": round(value, 3), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if wl == "cfg4_lens_f32" else "f64", "data": "synthetic",
        "config": {"workload": {
            "cfg2_planar": "BASELINE configs[1]: 128-element array, 1 planar interface (z=20 mm, c=2330/1483 m/s), "
                           "128x128 focal grid, fp64, per GPU",
            "cfg3_planar": "BASELINE configs[2]: 256-element array, 2 planar interfaces, 512x512 focal grid, fp64, per GPU",
            "cfg4_lens_f32": "BASELINE configs[3]: curved parametric interface (the reference lens), 128-element block of a "
                             "1024-element array x 1024x1024 target grid, fp32, per GPU",
            "cfg5_fmc": "BASELINE configs[4]: FMC tx/rx travel-time table, 3-layer medium + planar reflector, 256 tx rows x "
                        "2048 rx, per GPU",
            "ref_sweep": "reference sweep main_rt.py:464-501: 210 geometries x 905 rays forward trace + 65-element matcher",
            "ref_scale": "reference geometry, 1024 tx x 8192 rays forward trace + 65-element matcher"}[wl],
            "solves_per_step_per_gpu": units_per_step,
            "numerics": {"cfg4_lens_f32": "fp32 throughout (|dt| < 2e-10 s vs fp64)"}.get(
                wl, "fp64 results (planar solver: max |dt| 3.3e-17 s vs the long-double oracle); its Newton PRE-iteration runs "
                    "on the fp32 pipe, the result comes from an fp64 evaluation + Fermat expansion"
                if wl in ("cfg2_planar", "cfg3_planar", "cfg5_fmc") else "fp64, reference-compatible arithmetic"),
            "sharding": f"tx-element rows x{world}" + (", RCCL all-gather every step (overlapped)" if gather else
                                                       ", RCCL all-gather of the final matrix inside the timed region" if gather_end else
                                                       ", RCCL all-gather of the final matrix after the timed region" if gather_after else ""),
            "launch": (f"hipGraph replay of K launches on {args.streams} stream(s)" if args.streams > 1 else "hipGraph replay of K launches") +
                      f" (after ~{PREWARM_MS:.0f} ms of untimed replays: clock ramp)" if graph is not None
                      else "eager launches",
        },
    }
    if reassembly_ms is not None:
        shard_bytes = units_per_step * (4 if wl == "cfg4_lens_f32" else 8)
        out["reassembly"] = {"collective": "all_gather_into_tensor (RCCL)", "ms": round(reassembly_ms, 4),
                             "bytes_received_per_rank": shard_bytes * (world - 1),
                             "GB_per_s_per_rank": round(shard_bytes * (world - 1) / (reassembly_ms * 1e-3) / 1e9, 2)}
    if gather_error is not None:
        out["reassembly"] = {"collective": "all_gather_into_tensor (RCCL)", "error": gather_error}
    ach = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic_r01.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get(wl)
        except Exception:
            traffic = None
    out["roofline"] = {
        "bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(kern_ms, 5),
        "note": "scalar root-find: 4-16 B of HBM traffic per solve, so the HBM fraction is small by "
                "construction; the binding resource is VALU issue (see DESIGN.md, profiles/)",
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl)
    if rank == 0 and world == 1 and not args.no_extra and wl == "cfg2_planar":
        out["extra"] = extra_ref_path(dev_api, rtus, t64, torch)
        if not args.no_cpu_baseline:
            out["extra"]["cpu_ref_path"] = cpu_ref_path()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(wl):
    """Oracle-side CPU port timed on this host (bounded sample of the same workload)."""
    from oracle import cport
    cores = cport.num_threads()
    if wl in ("cfg2_planar", "cfg3_planar"):
        W = planar_inputs(wl, 0, 1)
        ne = min(W["n_e"], 128 if wl == "cfg2_planar" else 4)      # cfg2: the whole workload per call (2.1 M solves)
        cport.tt_layers_newton(W["z_if"], W["c"], W["xe"][:1], W["ze"][:1], W["xf"][:1024], W["zf"][:1024])
        reps, t0 = 0, time.perf_counter()
        while True:
            cport.tt_layers_newton(W["z_if"], W["c"], W["xe"][:ne], W["ze"][:ne], W["xf"], W["zf"])
            reps += 1
            if time.perf_counter() - t0 > 10.0:
                break
        dt = time.perf_counter() - t0
        v = reps * ne * W["n_f"] / dt / 1e6
        return {"value": round(v, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x ({ne} elements x {W['n_f']} focal points) of the same workload, "
                          f"oracle/rt_oracle.c orc_tt_layers_newton (fp64 Newton, OpenMP)"}
    if wl == "cfg5_fmc":
        W = fmc_inputs(0, 1)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0:
            cport.tt_layers_newton(W["z_if"], W["c"], W["xe"][:64], W["ze"][:64], W["xf"], W["zf"])
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": round(reps * 64 * W["n_f"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x (64 tx x 2048 rx) of the same table, orc_tt_layers_newton (fp64, OpenMP)"}
    if wl == "cfg4_lens_f32":
        W = lens_inputs(0, 1)
        import rtus
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0:
            cport.tt_lens(W["xe"][:8], W["ze"][:8], W["xf"][:65536], W["zf"][:65536], -rtus.ALPHA_MAX, rtus.ALPHA_MAX)
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": round(reps * 8 * 65536 / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x (8 elements x 65536 targets), orc_tt_lens (golden-section search in long double — a "
                          "checker, not a tuned CPU solver; OpenMP)"}
    R = ref_inputs(wl)
    g = R["geoms"][:: max(1, R["geoms"].shape[0] // 16)][:16]
    xa = R["xa"][:8]
    reps, t0 = 0, time.perf_counter()
    while True:
        cport.shoot_batch(xa, R["za"][:xa.size], R["zf"], R["alpha"], g)
        reps += 1
        if time.perf_counter() - t0 > 10.0:
            break
    dt = time.perf_counter() - t0
    v = reps * g.shape[0] * xa.size * R["n"] / dt / 1e6
    return {"value": round(v, 5), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x ({g.shape[0]} geometries x {xa.size} tx x {R['n']} rays) forward trace, "
                      f"oracle/rt_oracle.c orc_shoot_batch (fp64, O(N) polyline scan per ray, OpenMP)"}


def cpu_ref_path():
    """CPU legs of the reference-parity path on this host: the NumPy port (the reference's own O(N^2) scan as
    2-D array ops, 1 core) and the C port (OpenMP) — forward trace only, reference sweep geometry."""
    from oracle import cport, rt_numpy
    R = ref_inputs("ref_sweep")
    d = float(R["za"][0])
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 4.0:
        g = R["geoms"][k % len(R["geoms"])]
        rt_numpy.shoot(0.0, d, R["zf"], R["alpha"], g[0], g[1])
        k += 1
    np_rate = k * R["n"] / (time.perf_counter() - t0)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 4.0:
        cport.shoot_batch(R["xa"], R["za"], R["zf"], R["alpha"], R["geoms"])
        k += 1
    c_rate = k * R["geoms"].shape[0] * R["n"] / (time.perf_counter() - t0)
    return {"numpy_port_rays_per_s": round(np_rate, 1), "numpy_port_cores": 1,
            "c_port_rays_per_s": round(c_rate, 1), "c_port_cores": cport.num_threads(),
            "note": "forward trace at N = 905 (reference measured in SURVEY: 5.3 k rays/s on one core)"}


def _best_ms(torch, fn, k, blocks=5):
    """min over `blocks` of the mean time of k back-to-back calls (ms): a one-off host stall does not end up in the figure"""
    best = float("inf")
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / k * 1e3)
    return best


def extra_ref_path(dev_api, rtus, t64, torch):
    """Side measurement (not the headline): the reference-parity path on the same GPU."""
    res = {}
    for kind, fast in (("ref_sweep", False), ("ref_scale", False), ("ref_scale", True)):
        R = ref_inputs(kind)
        G, T, N = R["geoms"].shape[0], R["xa"].size, R["n"]
        plan = dev_api.ShootPlan(G, T, N, want=("tof", "land_x"), params=rtus.Params(), fast=fast)
        a = [t64(R[k]) for k in ("geoms", "xa", "za", "alpha", "zf")]
        x_rx = t64(R["x_rx"])
        box = [None]

        def one_pass():
            o = plan.run(*a)
            box[0] = dev_api.match_dev(o["land_x"].view(G * T, N), o["tof"].view(G * T, N), x_rx, out=box[0])

        for _ in range(3):
            one_pass()
        dt = _best_ms(torch, one_pass, 20 if kind == "ref_sweep" else 5) * 1e-3
        res[kind + ("_fastmath" if fast else "")] = {"Mrays_per_s": round(G * T * N / dt / 1e6, 2), "ms_per_pass": round(dt * 1e3, 4),
                     "rays_per_pass": G * T * N}
    # the same planar kernel on BASELINE configs[2] (256 elements, 2 interfaces, 512 x 512 grid): a launch large
    # enough (537 MB of results) that launch overhead and ramp no longer weigh on the HBM fraction
    W = planar_inputs("cfg3_planar", 0, 1)
    out3 = torch.empty((W["n_e"], W["n_f"]), dtype=torch.float64, device="cuda")
    plan3 = dev_api.LayersPlan(W["z_if"], W["c"], t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"]), out=out3)
    for _ in range(3):
        plan3.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan3.run()
    e1.record()
    torch.cuda.synchronize()
    ms3 = e0.elapsed_time(e1) / 20
    n3 = W["n_e"] * W["n_f"]
    res["cfg3_planar"] = {"Mrays_per_s": round(n3 / ms3 / 1e3, 1), "ms_per_launch": round(ms3, 4), "solves_per_launch": n3,
                          "hbm_frac": round(n3 * 8 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del out3, plan3
    # curved-lens two-point Fermat solves (BASELINE config 4 geometry: 1024 elements over the reference lens,
    # 1024 x 256 target strip inside the insonified cone), fp64 and fp32
    import ctypes as C
    L = rtus.lib()
    lens = rtus.Params().lens()
    d = rtus.Params().d
    xe = (np.arange(1024) - 511.5) * 0.3e-4
    xs, zs = np.meshgrid(np.linspace(-0.004, 0.004, 1024), np.linspace(0.03, 0.07, 256))
    for name, dt, fn in (("lens_fermat_f64", torch.float64, L.rtus_tt_lens_dev), ("lens_fermat_f32", torch.float32, L.rtus_tt_lens_f32_dev)):
        mk = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda").to(dt).contiguous()
        txe, tze, txf, tzf = mk(xe), mk(np.full(1024, d)), mk(xs.ravel()), mk(zs.ravel())
        out = torch.empty((1024, txf.numel()), dtype=dt, device="cuda")
        run = lambda: fn(C.byref(lens), -rtus.ALPHA_MAX, rtus.ALPHA_MAX, txe.data_ptr(), tze.data_ptr(), 1024,
                         txf.data_ptr(), tzf.data_ptr(), txf.numel(), out.data_ptr(), None,
                         torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            assert run() == 0
        dtm = _best_ms(torch, run, 3, blocks=3) * 1e-3
        n = 1024 * txf.numel()
        res[name] = {"Mrays_per_s": round(n / dtm / 1e6, 1), "ms_per_pass": round(dtm * 1e3, 3), "solves_per_pass": n}
    # root-finding pulse-echo solve: the reference sweep's 210 geometries x 65 elements
    R = ref_inputs("ref_sweep")
    sol = [None]

    def one_solve():
        sol[0] = rtus.solve_travel_times(R["xa"], R["za"], R["x_rx"], R["alpha"], R["geoms"], params=rtus.Params())[0]

    dtm = _best_ms(torch, one_solve, 3, blocks=3) * 1e-3
    tt = sol[0]
    res["solve_sweep_host_api"] = {"ms_per_pass": round(dtm * 1e3, 3), "elements": int(tt.size),
                                   "with_root": int(np.isfinite(tt).sum()), "note": "host-buffer API incl. PCIe + alloc"}
    return res


if __name__ == "__main__":
    main()

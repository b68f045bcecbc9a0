#!/usr/bin/env python3
"""bench.py — Mrays/s (element x focal-point travel-time solves per second) on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM.

N = 1 (default): BASELINE.json configs[2], the largest single-GPU configuration — 256-element array, 2 planar
interfaces, 512 x 512 focal grid, fp64 = 67,108,864 Fermat travel-time solves (537 MB of results) per step
(rtus_tt_layers_dev, csrc/rtus_fermat.hip).  The K timed steps are K back-to-back launches on one stream, captured once
into a hipGraph and replayed (--graph off: eager launches); the figure does not depend on K (20 steps = 3.6 ms of GPU
work against ~20 us of launch + sync latency).

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): BASELINE.json configs[3] STRONG-scaled — the
1024-element x 1024 x 1024-target curved-lens table (fp32) with its tx rows sharded 1024/N per GPU.  The solves need no
exchange, so the timed region is collective-free (`value`); the one exchange of the path, the all-gather that
reassembles the table on every rank, is timed on its own right after and folded into `value_with_reassembly`
(K steps + ONE all-gather, what --gather end would time) and `value_reassembled_every_step`.  `extra.cfg5_fmc` does the
same for configs[4] (2048 x 2048 FMC table, tx rows sharded).  Planar workloads given explicitly (--workload
cfg2_planar / cfg3_planar) are weak-scaled instead: every rank solves its own element block of an N-times wider aperture.

Other workloads (--workload): cfg2_planar (configs[1]), cfg4_lens_f32, cfg5_fmc, ref_sweep (the reference's own sweep,
main_rt.py:464-501: 210 geometries x 905 rays forward trace + 65-element matcher), ref_scale (reference geometry,
1024 tx x 8192 rays).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against the HBM roof as the contract asks
(algorithmic bytes per launch / HIP-event launch time; `traffic` = PMC-measured HBM bytes per launch from profiles/);
`roofline_valu` carries what actually binds this scalar root-find (VALU issue; SQ counters from profiles/);
`cpu_baseline` is the oracle's fp64 CPU port timed on this box's host cores on the whole workload, and `accuracy` the
measured max |dt| of the GPU result against the long-double oracle on a seeded sample (rank 0, N = 1 only);
`extra` holds side measurements of the other configs and of the reference-parity path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREWARM_MS = 300.0             # untimed graph replays before the timed region (clock ramp), disclosed in config.  Measured on the
                               # headline (scripts/exp_bench_prewarm.sh, K = 20, three rounds per value, two boxes): 0 ms 0.185 ms per step,
                               # 3 ms 0.163, 30 ms 0.1385 / 0.1334, 300 ms 0.1352 / 0.1327, 1000 ms 0.139 - 0.149 (the power cap answers):
                               # an MI355X leaving idle needs a few hundred ms of load to reach the clock it then holds
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (spec)
PLANAR = ("cfg2_planar", "cfg3_planar", "cfg5_fmc")
WORKLOADS = ("cfg2_planar", "cfg3_planar", "cfg4_lens_f32", "cfg5_fmc", "ref_sweep", "ref_scale")
DESCR = {
    "cfg2_planar": "BASELINE configs[1]: 128-element array, 1 planar interface (z=20 mm, c=2330/1483 m/s), 128x128 focal "
                   "grid, fp64",
    "cfg3_planar": "BASELINE configs[2]: 256-element array, 2 planar interfaces (nested root-find), 512x512 focal grid, fp64",
    "cfg4_lens_f32": "BASELINE configs[3]: 1024-element phased array, curved parametric interface (the reference lens), "
                     "1024x1024 target grid, fp32, tx rows sharded across the GPUs",
    "cfg5_fmc": "BASELINE configs[4]: full-matrix-capture 2048x2048 tx/rx travel-time table, 3-layer medium + planar "
                "reflector, tx rows sharded across the GPUs",
    "ref_sweep": "reference sweep main_rt.py:464-501: 210 geometries x 905 rays forward trace + 65-element matcher",
    "ref_scale": "reference geometry, 1024 tx x 8192 rays forward trace + 65-element matcher",
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto", choices=("auto",) + WORKLOADS,
                    help="auto: cfg3_planar on one GPU (BASELINE configs[2], the largest single-GPU config), "
                         "cfg4_lens_f32 strong-scaled on several (BASELINE configs[3])")
    ap.add_argument("--gather", default="after", choices=["after", "end", "step", "off"],
                    help="--gpus > 1: RCCL all-gather of the row shards — once AFTER the timed region, timed on its own and "
                         "folded into value_with_reassembly (after, default); once inside the timed region after the K "
                         "steps (end); after every step, overlapped with the next kernel (step); never (off)")
    ap.add_argument("--graph", default="on", choices=["on", "off"],
                    help="replay the K timed steps as one captured hipGraph (N=1 or --gather after/end/off)")
    ap.add_argument("--streams", type=int, default=1,
                    help="planar workloads, graph mode: capture the K steps round-robin on this many streams, each with its "
                         "own output buffer (default 1: K back-to-back launches on one stream)")
    ap.add_argument("--tier", default="taup", choices=["taup", "accurate"],
                    help="planar workloads: accuracy tier of the solver — taup (RTUS_TT_TAUP_TAIL: the travel time from the tau-p form, "
                         "<= 6e-11 relative at worst; the measured max |dt| is in the line) or accurate (the library's default tier)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements")
    args = ap.parse_args(argv)
    if args.workload == "auto":
        args.workload = "cfg3_planar" if args.gpus <= 1 else "cfg4_lens_f32"
    return args


# ------------------------------------------------------------------------------------ workloads
def scaling_of(wl, world):
    """configs[3] / configs[4] are BASELINE's multi-GPU shapes: a fixed table whose tx rows are sharded (strong scaling).
    The single-GPU planar configs, run on several GPUs, keep their per-GPU block (weak scaling)."""
    return "strong" if wl in ("cfg4_lens_f32", "cfg5_fmc") else "weak"


# RTUS_BENCH_SMALL=1 (rehearsals of the N > 1 control flow on a one-GPU box, tests/test_gpu_dist_two_ranks.py): the same
# workloads with a few rows and a coarse grid.  Never set for a measurement: the line says so in config.workload.
SMALL = os.environ.get("RTUS_BENCH_SMALL", "0") == "1"


def planar_inputs(cfg, rank, world):
    """cfg2 = 128 elems @0.6 mm on z=0, interface z=20 mm, c=(2330,1483), 128x128
    focal grid over x in [-20,20] mm, z in [25,65] mm; cfg3 = 256 elems @0.3 mm, interfaces at
    10/25 mm, c=(2330,1483,5900), 512x512 grid.  Rank r owns elements [r*n_e, (r+1)*n_e) of a
    world*n_e-element aperture (weak scaling)."""
    if cfg == "cfg2_planar":
        n_e, pitch, z_if, c, g, zr = 128, 0.6e-3, [0.020], [2330.0, 1483.0], 128, (0.025, 0.065)
    else:
        n_e, pitch, z_if, c, g, zr = 256, 0.3e-3, [0.010, 0.025], [2330.0, 1483.0, 5900.0], 512, (0.026, 0.066)
    if SMALL:
        n_e, g = 24, 96
    n_all = n_e * world
    x_all = (np.arange(n_all) - (n_all - 1) / 2.0) * pitch
    xe = x_all[rank * n_e:(rank + 1) * n_e]
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, g), np.linspace(zr[0], zr[1], g))
    return dict(z_if=z_if, c=c, xe=xe, ze=np.zeros(n_e), xf=xs.ravel(), zf=zs.ravel(), n_e=n_e, n_f=g * g,
                rows_total=n_all, lo=rank * n_e)


def _strong_rows(n_rows, rank, world, n_f, elem_bytes):
    """rows [lo, hi) of rank `rank`: equal shards rounded up to the table kernels' rows per workgroup, so that every shard
    starts on a block boundary of the whole table and the sharded table is bit for bit the one-GPU table (rtus_shard_rows)"""
    import rtus
    per = int(rtus.lib().rtus_shard_rows(n_rows, n_f, elem_bytes, world))
    assert per > 0
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows), per


def lens_inputs(rank, world, n_rows=1024):
    """BASELINE configs[3]: 1024-element array @0.03 mm on z = d over the reference lens h(alpha),
    water below, 1024 x 1024 target grid inside the insonified cone, fp32; tx rows sharded 1024/world per GPU (strong
    scaling: the table is fixed, each rank solves rows [lo, hi))."""
    import rtus
    g = 1024
    if SMALL:
        n_rows, g = min(n_rows, 64), 128
    x_all = (np.arange(n_rows) - (n_rows - 1) / 2.0) * 0.3e-4
    lo, hi, per = _strong_rows(n_rows, rank, world, g * g, 4)
    xs, zs = np.meshgrid(np.linspace(-0.004, 0.004, g), np.linspace(0.03, 0.07, g))
    return dict(xe=x_all[lo:hi], ze=np.full(hi - lo, rtus.Params().d), xf=xs.ravel(), zf=zs.ravel(),
                n_e=hi - lo, n_f=xs.size, rows_total=n_rows, lo=lo)


def fmc_inputs(rank, world, n_tx=2048):
    """BASELINE configs[4]: 2048 tx x 2048 rx on z = 0, 3 horizontal layers, planar reflector at
    depth — unfolded about the reflector into a 5-layer one-way problem; tx rows sharded 2048/world per GPU (strong)."""
    z_if, c, z_r = np.array([0.008, 0.020]), np.array([2330.0, 1483.0, 5900.0]), 0.035
    z_m = np.concatenate([z_if, (2.0 * z_r - z_if)[::-1]])
    c_m = np.concatenate([c, c[::-1][1:]])
    if SMALL:
        n_tx = min(n_tx, 64)
    x_tx = (np.arange(n_tx) - (n_tx - 1) / 2.0) * 0.3e-3
    x_rx = (np.arange(2048) - 1023.5) * 0.3e-3
    lo, hi, per = _strong_rows(n_tx, rank, world, 2048, 8)
    return dict(z_if=z_m, c=c_m, xe=x_tx[lo:hi], ze=np.zeros(hi - lo), xf=x_rx, zf=np.full(2048, 2.0 * z_r),
                n_e=hi - lo, n_f=2048, rows_total=n_tx, lo=lo)


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


def workload_inputs(wl, rank, world):
    if wl == "cfg4_lens_f32":
        return lens_inputs(rank, world)
    if wl == "cfg5_fmc":
        return fmc_inputs(rank, world)
    return planar_inputs(wl, rank, world)


# ------------------------------------------------------------------------------------ timing helpers
class Timed:
    """K steps of `step(s)` as one hipGraph (or eagerly), bracketed as the contract says."""

    def __init__(self, torch, step, steps, streams=1, use_graph=True):
        self.torch, self.step, self.steps = torch, step, steps
        self.graph = None
        if use_graph:
            try:
                self.graph = self._capture(streams)
            except Exception as exc:            # capture unsupported -> eager launches
                print(f"[bench] hipGraph capture failed ({exc!r}); timing eager launches", file=sys.stderr)
                self.graph = None
                torch.cuda.synchronize()

    def _capture(self, streams):
        torch = self.torch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            # thread_local: other threads (e.g. the RCCL watchdog at --gpus > 1) may keep issuing HIP calls
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                if streams > 1:
                    subs = [torch.cuda.Stream() for _ in range(streams)]              # fork ...
                    for st in subs:
                        st.wait_stream(side)
                    for s in range(self.steps):
                        with torch.cuda.stream(subs[s % streams]):                    # step s -> stream / buffer s % S
                            self.step(s)
                    for st in subs:                                                   # ... and join
                        side.wait_stream(st)
                else:
                    for s in range(self.steps):
                        self.step(s)
        torch.cuda.current_stream().wait_stream(side)
        # untimed replays: the first uploads the graph; then keep the GPU busy for PREWARM_MS so that the timed region
        # starts at the sustained clock whatever K is (an idle MI355X needs a few hundred ms of load to bring its clocks up)
        t_pre = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        one = max(time.perf_counter() - t_pre, 1e-5)
        for _ in range(min(max(1, 4000 // self.steps), int(PREWARM_MS * 1e-3 / one))):     # (at most ~4,000 launches: short kernels ramp less)
            g.replay()
        torch.cuda.synchronize()
        return g

    def run(self, barrier, tail=None):
        """-> (wall seconds of the bracketed region, mean launch ms from HIP events on the launch stream)."""
        torch = self.torch
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); ev1.record()        # a torch event creates its hipEvent at the first record: 30-70 us of host time that would
                                          # otherwise sit between t0 and the first launch (scripts/exp_bench_region.sh: 2.5 % of a K = 20 region)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()                      # same (current) stream the library launches on
        if self.graph is not None:
            self.graph.replay()
        else:
            for s in range(self.steps):
                self.step(s)
        ev1.record()
        if tail is not None:
            tail()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        return dt, ev0.elapsed_time(ev1) / self.steps


def load_profile_json(name):
    """profiles/<name>: the newest round's file wins (r04 over r03 over ...)."""
    for tag in ("r04", "r03", "r02", "r01"):
        p = os.path.join(ROOT, "profiles", name.format(tag=tag))
        if os.path.exists(p):
            try:
                return json.load(open(p)), os.path.relpath(p, ROOT)
            except Exception:
                pass
    return None, None


# ------------------------------------------------------------------------------------ main
def main(argv=None):
    args = parse(argv)
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
    dist_api = import_module("ray-tracing-ultrasound_amd.dist")

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

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    wl = args.workload
    scaling = scaling_of(wl, world)
    gather_step = world > 1 and args.gather == "step"
    gather_end = world > 1 and args.gather == "end"
    gather_after = world > 1 and args.gather == "after"

    # ---- the step ---------------------------------------------------------------------------------
    m = None                                  # RowShardedMatrix of the table workloads
    if wl in PLANAR or wl == "cfg4_lens_f32":
        W = workload_inputs(wl, rank, world)
        n_e, n_f = W["n_e"], W["n_f"]
        f32 = wl == "cfg4_lens_f32"
        tdt = torch.float32 if f32 else torch.float64
        slots = 2 if gather_step else (max(1, args.streams) if wl in PLANAR else 1)
        align = dev_api.rows_per_block(W["rows_total"], n_f, tdt) if scaling == "strong" else 1
        m = dist_api.RowShardedMatrix(W["rows_total"], n_f, dtype=tdt, device=dev, slots=slots, align=align)
        if m.per < n_e or (scaling == "strong" and (m.lo, m.hi) != (W["lo"], W["lo"] + n_e)):
            raise SystemExit("row sharding of the inputs and of the matrix disagree")
        units_per_step = n_e * n_f                       # this rank's solves per step
        total_units_per_step = W["rows_total"] * n_f     # all ranks
        word = 4 if f32 else 8
        alg_bytes = units_per_step * word + (2 * n_e + 2 * n_f) * word
        if f32:
            import ctypes as C
            t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
            xe, ze, xf, zf = t32(W["xe"]), t32(W["ze"]), t32(W["xf"]), t32(W["zf"])
            lens = rtus.Params().lens()
            fn = rtus.lib().rtus_tt_lens_f32_rows_dev        # rows [lo, lo + n_e) of the rows_total-row table: the one-GPU table's bits
            kernel = "rtus_tt_lens_kernel<float, true, false>"

            def launch(b):
                st = fn(C.byref(lens), -rtus.ALPHA_MAX, rtus.ALPHA_MAX, xe.data_ptr(), ze.data_ptr(), n_e, W["lo"], W["rows_total"],
                        xf.data_ptr(), zf.data_ptr(), n_f, m.local(b).data_ptr(), None, torch.cuda.current_stream().cuda_stream)
                assert st == 0
        else:
            xe, ze, xf, zf = t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"])
            taup = args.tier == "taup"
            plans = [dev_api.LayersPlan(W["z_if"], W["c"], xe, ze, xf, zf, out=m.local(b)[:n_e], taup=taup) for b in range(slots)]
            kernel = f"rtus_tt_layers_kernel<{len(W['c'])}, false, {'true' if taup else 'false'}, false>"

            def launch(b):
                plans[b].run()

        def step(s):
            b = s % slots
            if gather_step:
                m.wait(b)               # the slot's previous all-gather must finish before it is rewritten
            launch(b)
            if gather_step:
                m.gather(b, async_op=True)      # RCCL, overlaps the next step's kernel

        def drain():
            for b in range(slots):
                m.wait(b)

        def reassemble():               # the one exchange of the path: RCCL all-gather of the row blocks over xGMI
            m.gather(0)
    else:
        R = ref_inputs(wl)
        G, T, N = R["geoms"].shape[0], R["xa"].size, R["n"]
        # one body of the reference's parameter loop for every row: forward trace + element matcher in ONE kernel (rtus_sweep_dev)
        plan = dev_api.SweepPlan(G, T, N, R["x_rx"].size, want=("tof", "land_x"), params=rtus.Params(), atol=1e-6, rtol=1e-5, device=dev)
        geoms, xa, za, alpha, zf = t64(R["geoms"]), t64(R["xa"]), t64(R["za"]), t64(R["alpha"]), t64(R["zf"])
        x_rx = t64(R["x_rx"])
        units_per_step = G * T * N
        total_units_per_step = units_per_step * world
        alg_bytes = units_per_step * 16 + N * 16       # tof + land_x written per ray; alpha, z_f read once
        kernel = "rtus_shoot_kernel<false, false, true>"

        def step(s):
            plan.run(geoms, xa, za, alpha, zf, x_rx)

        def drain():
            pass

        def reassemble():
            pass

        gather_step = gather_end = gather_after = False     # independent rows per rank, no table to reassemble

    # ---- warm-up ----------------------------------------------------------------------------------
    for s in range(args.warmup):
        step(s)
    gather_error = None
    if world > 1 and args.gather != "off":
        try:
            reassemble()                # also warms the communicator up
        except Exception as exc:        # the timed region needs no collective: keep measuring, report the failure
            if not gather_after:
                raise
            gather_error = repr(exc)
            print(f"[bench] warm-up all-gather failed: {gather_error}", file=sys.stderr)
    drain()
    torch.cuda.synchronize()

    # ---- timed region: K steps, barrier + synchronize on both sides, max over ranks -----------------
    # Python + hipLaunchKernel cost ~10 us per call, so the K launches are captured once into a hipGraph and replayed
    # (same kernels, same order, no per-launch host work).
    timed = Timed(torch, step, args.steps, streams=args.streams if wl in PLANAR else 1,
                  use_graph=args.graph == "on" and not gather_step)

    def tail():
        if gather_end:
            reassemble()
        drain()

    dt, kern_ms = timed.run(barrier, tail)
    dt = max_over_ranks(dt)
    reassembly_ms = None
    if gather_after and gather_error is None:         # the one exchange of the path, outside the solve loop
        try:
            barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reassemble()
            torch.cuda.synchronize()
            barrier()
            reassembly_ms = max_over_ranks((time.perf_counter() - t1) * 1e3)
        except Exception as exc:
            gather_error = repr(exc)
            print(f"[bench] all-gather failed: {gather_error}", file=sys.stderr)

    total_units = total_units_per_step * args.steps
    value = total_units / dt / 1e6
    graph_on = timed.graph is not None
    out = {
        "metric": "Mrays/sec (elem x focal travel-time solves)", "value": round(value, 3), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 6), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32" if wl == "cfg4_lens_f32" else "f64", "data": "synthetic",
        "config": {
            "workload": ("REHEARSAL SIZES (RTUS_BENCH_SMALL=1), not a measurement — " if SMALL else "") + DESCR[wl] + ("" if world == 1 else
                                     f"; {scaling}-scaled over {world} GPUs" +
                                     (f" ({units_per_step // n_f if m is not None else 0} tx rows per GPU)" if m is not None else "")),
            "tier": ("RTUS_TT_TAUP_TAIL (rtus_tt_layers_ex*, taup=True; <= 6e-11 relative)" if args.tier == "taup" else
                     "default (flags = 0; <= 1e-13 relative)") if wl in PLANAR else "n/a",
            "solves_per_step_per_gpu": units_per_step,
            "solves_per_step_all_gpus": total_units_per_step,
            "numerics": {"cfg4_lens_f32": "fp32 throughout (|dt| < 2e-10 s vs fp64 where both follow the same ray path; DESIGN section 4 on minima at the end of the search interval)"}.get(
                wl, ("fp64 results, tier RTUS_TT_TAUP_TAIL: Newton PRE-iteration on the fp32 pipe, the travel time from the tau-p form "
                     "p X + sum (h/c) cos(theta) in fp64 (stationary in p) + its second-order term from the fp32 residual — <= 6e-11 "
                     "relative by construction, measured max |dt| vs the long-double oracle: see `accuracy`; the library's default tier "
                     "(fp64 residual, <= 1e-13 relative) is timed in extra.cfg3_planar_accurate_tier" if args.tier == "taup" else
                     "fp64 results, default tier: Newton PRE-iteration on the fp32 pipe, the result from an fp64 evaluation + Fermat "
                     "expansion (measured max |dt| vs the long-double oracle: see `accuracy`)")
                if wl in PLANAR else "fp64, reference-compatible arithmetic"),
            "sharding": (f"tx-element rows x{world}" if m is not None else f"{world} rank(s), each its own (geometry, tx) rows: no exchange") + (", RCCL all-gather every step (overlapped)" if gather_step else
                                                       ", RCCL all-gather of the final matrix inside the timed region" if gather_end else
                                                       ", collective-free timed region; RCCL all-gather of the table timed right after" if gather_after else ""),
            "launch": ((f"hipGraph replay of K launches on {args.streams} stream(s)" if args.streams > 1 else "hipGraph replay of K launches") +
                       f" (after ~{PREWARM_MS:.0f} ms of untimed replays: clock ramp)") if graph_on else "eager launches",
        },
    }
    if reassembly_ms is not None:
        shard_bytes = m.per * n_f * word
        per_step_s = dt / args.steps
        out["reassembly"] = {"collective": "all_gather_into_tensor (RCCL over xGMI)" if backend == "nccl" else f"all_gather_into_tensor ({backend})",
                             "ms": round(reassembly_ms, 4),
                             "bytes_received_per_rank": shard_bytes * (world - 1),
                             "GB_per_s_received_per_rank": round(shard_bytes * (world - 1) / (reassembly_ms * 1e-3) / 1e9, 2)}
        out["value_with_reassembly"] = round(total_units / (dt + reassembly_ms * 1e-3) / 1e6, 3)     # K steps + ONE all-gather
        out["value_reassembled_every_step"] = round(total_units_per_step / (per_step_s + reassembly_ms * 1e-3) / 1e6, 3)
    if gather_error is not None:
        out["reassembly"] = {"collective": "all_gather_into_tensor (RCCL)", "error": gather_error}
    if world > 1:
        # what `value` is, where nobody can overlook it: the solves alone; the table's reassembly is timed beside it
        out["value_basis"] = ("collective-free" if (gather_after or args.gather == "off") else
                              "with the all-gather of the final table" if gather_end else "with an all-gather every step")
        rs = out.get("value_with_reassembly")
        out["config"]["numerics"] = (f"VALUE = {out['value']} Mrays/s is the COLLECTIVE-FREE solve rate; with ONE RCCL all-gather of the table after the "
                                     f"K steps: {rs} Mrays/s (value_with_reassembly), reassembled every step: {out.get('value_reassembled_every_step')} "
                                     f"Mrays/s — the table's reassembly, not the solve, decides those (DESIGN.md section 6); the workload that "
                                     f"scales WITH its collective inside the timed region is extra.tfm_columns.  " + out["config"]["numerics"])
    if world > 1:
        out["multi_gpu_note"] = ("RCCL over xGMI first ran in the driver's scaling job (the build pool has one GPU per box); compare "
                                 "this line with `extra.cfg4_lens_f32_full` (the same table on one GPU) of the N = 1 line" if wl == "cfg4_lens_f32"
                                 else "per-GPU work fixed (weak scaling): no data-path collective")

    ach = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic_doc, traffic_src = load_profile_json("traffic_{tag}.json")
    traffic = (traffic_doc or {}).get(wl)
    out["roofline"] = {
        "bound": "hbm", "binds": "valu_issue", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(kern_ms, 5),
        "note": "HBM fraction as the contract asks (8 B written per fp64 solve, 4 B per fp32 solve); a scalar root-find moves "
                "4-16 B per solve, so the binding resource is VALU issue, not HBM: see roofline_valu",
    }
    valu_doc, valu_src = load_profile_json("valu_{tag}.json")
    if valu_doc and wl in valu_doc:
        v = dict(valu_doc[wl])
        v["source"] = valu_src + " (rocprofv3 --pmc SQ_* pass of this workload; not collected in this run)"
        if v.get("issue_cycles_per_wave_solve") and v.get("clock_ghz"):
            # what the measured instruction mix would take at 100 % VALU issue efficiency vs what the launch took
            simd = 1024.0
            t_issue = (units_per_step / 64.0 / simd) * v["issue_cycles_per_wave_solve"] / (v["clock_ghz"] * 1e9)
            v["issue_bound_ms"] = round(t_issue * 1e3, 5)
            v["frac_of_issue_bound"] = round(t_issue * 1e3 / kern_ms, 4)
        out["roofline_valu"] = v

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        gpu_table = m.local(0)[:n_e] if m is not None else None
        out["cpu_baseline"], acc = cpu_baseline(wl, gpu_table, torch)
        if acc is not None:
            out["accuracy"] = acc
            out["max_abs_dt_s"] = acc["max_abs_dt_s"]
    if rank == 0 and world == 1 and not args.no_extra and wl == "cfg3_planar":
        out["extra"] = extra_measurements(dev_api, dist_api, rtus, t64, torch, dev)
        lens_samples = out["extra"].pop("_lens_samples", {})
        if not args.no_cpu_baseline:
            out["extra"]["cpu_ref_path"] = cpu_ref_path()
            out["extra"]["lens_tables_accuracy"] = lens_tables_accuracy(lens_samples)
    if world > 1 and not args.no_extra and wl == "cfg4_lens_f32" and args.gather != "step":
        fm = fmc_strong(dev_api, dist_api, t64, torch, dev, rank, world, barrier, max_over_ranks, args)
        tc = tfm_columns(dev_api, dist_api, t64, torch, dev, rank, world, barrier, max_over_ranks, args)
        if rank == 0:
            out["extra"] = {"cfg5_fmc": fm, "tfm_columns": tc}
    if world == 1 and rank == 0 and not args.no_extra and "extra" in out:
        out["extra"]["tfm_columns"] = tfm_columns(dev_api, dist_api, t64, torch, dev, 0, 1, barrier, max_over_ranks, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def tfm_columns(dev_api, dist_api, t64, torch, dev, rank, world, barrier, max_over_ranks, args):
    """The pipeline that needs every element per focal point — travel-time table -> TFM delay-and-sum — sharded by COLUMNS (focal
    points), strong-scaled: rank r solves tt[all 64 elements, its slice of a 1024 x 1024 image], beamforms the slice, and the image
    slices (4 B per focal point) are all-gathered INSIDE the timed region every step (dist.tfm_layers_sharded).  No travel time
    crosses a link: this is the multi-GPU shape of the path whose collective is small enough to scale with."""
    import rtus  # noqa: F401
    g = 256 if SMALL else 1024
    n_el, n_t, fs = 64, 2048, 50e6
    z_if, c = [0.010, 0.025], [2330.0, 1483.0, 5900.0]
    xe = t64((np.arange(n_el) - (n_el - 1) / 2.0) * 0.6e-3)
    ze = t64(np.zeros(n_el))
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, g), np.linspace(0.026, 0.066, g))
    xf, zf = t64(xs.ravel()), t64(zs.ravel())
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    fmc = torch.randn((n_el, n_el, n_t), dtype=torch.float32, device=dev, generator=gen)
    res = {"workload": f"{n_el}-element FMC record ({n_el} x {n_el} x {n_t} samples, synthetic noise) -> {g} x {g} TFM image through 2 planar "
                       f"interfaces: travel-time table + delay-and-sum per focal-point slice, image slices all-gathered every step",
           "focal_points_per_gpu": -(-g * g // world)}
    try:
        step = lambda s=None: dist_api.tfm_layers_sharded(fmc, fs, z_if, c, xe, ze, xf, zf, t0=0.0)
        for _ in range(3):
            img = step()
        torch.cuda.synchronize()
        K = 10
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            img = step()
        torch.cuda.synchronize()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        pairs = float(n_el) * n_el * g * g
        res.update({"ms_per_image": round(dt / K * 1e3, 4), "steps": K, "G_pair_samples_per_s": round(pairs * K / dt / 1e9, 1),
                    "Mrays_per_s_table_part": round(n_el * g * g * K / dt / 1e6, 1),
                    "collective": "all_gather_into_tensor of the image slices inside the timed region" if world > 1 else "none (one GPU)",
                    "collective_bytes_received_per_rank": 4 * (g * g - g * g // world) if world > 1 else 0,
                    "image_checksum": float(img.double().abs().sum())})
    except Exception as exc:
        res["error"] = repr(exc)
    return res


def fmc_strong(dev_api, dist_api, t64, torch, dev, rank, world, barrier, max_over_ranks, args):
    """BASELINE configs[4] on N GPUs: the 2048 x 2048 FMC table, tx rows sharded, collective-free solve + ONE all-gather."""
    W = fmc_inputs(rank, world)
    n_e, n_f = W["n_e"], W["n_f"]
    m = dist_api.RowShardedMatrix(W["rows_total"], n_f, device=dev, slots=1, align=dev_api.rows_per_block(W["rows_total"], n_f))
    plan = dev_api.LayersPlan(W["z_if"], W["c"], t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"]), out=m.local(0)[:n_e],
                              row0=W["lo"], n_rows_total=W["rows_total"])
    for _ in range(5):
        plan.run()
    res = {"workload": DESCR["cfg5_fmc"], "rows_per_gpu": n_e}
    try:
        if args.gather != "off":
            m.gather(0)
        torch.cuda.synchronize()
        K = max(20, min(args.steps, 200))
        timed = Timed(torch, lambda s: plan.run(), K, use_graph=args.graph == "on")
        dt, kern_ms = timed.run(barrier)
        dt = max_over_ranks(dt)
        tot = W["rows_total"] * n_f
        res.update({"Mrays_per_s": round(tot * K / dt / 1e6, 1), "ms_per_step": round(dt / K * 1e3, 5), "steps": K,
                    "avg_launch_ms": round(kern_ms, 5)})
        if args.gather != "off":
            barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            m.gather(0)
            torch.cuda.synchronize()
            barrier()
            g_ms = max_over_ranks((time.perf_counter() - t1) * 1e3)
            res.update({"reassembly_ms": round(g_ms, 4),
                        "Mrays_per_s_reassembled_every_step": round(tot / (dt / K + g_ms * 1e-3) / 1e6, 1)})
    except Exception as exc:
        res["error"] = repr(exc)
    return res


# ------------------------------------------------------------------------------------ CPU legs (oracle = checker)
def cpu_baseline(wl, gpu_table, torch):
    """Oracle-side CPU port timed on this host + the GPU result checked against the long-double oracle on a seeded
    sample (the only place bench.py touches oracle/).  -> (cpu_baseline dict, accuracy dict or None)"""
    from oracle import cport
    cores = cport.num_threads()
    acc = None
    if wl in ("cfg2_planar", "cfg3_planar", "cfg5_fmc"):
        W = workload_inputs(wl, 0, 1)
        ne = W["n_e"]
        cport.tt_layers_newton(W["z_if"], W["c"], W["xe"][:1], W["ze"][:1], W["xf"][:1024], W["zf"][:1024])
        reps, t0 = 0, time.perf_counter()
        while True:                                                    # the WHOLE workload per call
            cport.tt_layers_newton(W["z_if"], W["c"], W["xe"], W["ze"], W["xf"], W["zf"])
            reps += 1
            if time.perf_counter() - t0 > 10.0:
                break
        dt = time.perf_counter() - t0
        base = {"value": round(reps * ne * W["n_f"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x the whole workload ({ne} elements x {W['n_f']} focal points), "
                          f"oracle/rt_oracle.c orc_tt_layers_newton (fp64 Newton, OpenMP)"}
        if gpu_table is not None:
            rng = np.random.default_rng(20260)
            re = np.sort(rng.choice(ne, size=min(16, ne), replace=False))
            cf = np.sort(rng.choice(W["n_f"], size=min(8192, W["n_f"]), replace=False))
            got = gpu_table[torch.as_tensor(re, device=gpu_table.device)][:, torch.as_tensor(cf, device=gpu_table.device)].cpu().numpy()
            ref = cport.tt_layers(W["z_if"], W["c"], W["xe"][re], W["ze"][re], W["xf"][cf], W["zf"][cf])
            same_nan = bool(np.array_equal(np.isnan(got), np.isnan(ref)))
            d = np.abs(got - ref)
            acc = {"max_abs_dt_s": float(np.nanmax(d)), "median_abs_dt_s": float(np.nanmedian(d)),
                   "max_rel": float(np.nanmax(d / np.abs(ref))), "nan_masks_identical": same_nan,
                   "sample": f"{re.size} seeded-random elements x {cf.size} seeded-random focal points of the timed workload's last step",
                   "checker": "oracle/rt_oracle.c orc_tt_layers (long-double bisection; pinned to 50-digit values, parity unpinned "
                              "by the reference: it has no planar interfaces)", "bar_s": 1e-9}
        return base, acc
    if wl == "cfg4_lens_f32":
        W = lens_inputs(0, 1)
        import rtus
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0:
            cport.tt_lens(W["xe"][:8], W["ze"][:8], W["xf"][:65536], W["zf"][:65536], -rtus.ALPHA_MAX, rtus.ALPHA_MAX)
            reps += 1
        dt = time.perf_counter() - t0
        base = {"value": round(reps * 8 * 65536 / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x (8 elements x 65536 targets), orc_tt_lens (golden-section search in long double — a "
                          "checker, not a tuned CPU solver; OpenMP)"}
        if gpu_table is not None:
            rng = np.random.default_rng(20261)
            re = np.sort(rng.choice(W["n_e"], size=8, replace=False))
            cf = np.sort(rng.choice(W["n_f"], size=2048, replace=False))
            got = gpu_table[torch.as_tensor(re, device=gpu_table.device)][:, torch.as_tensor(cf, device=gpu_table.device)].double().cpu().numpy()
            ref, _ = cport.tt_lens(W["xe"][re], W["ze"][re], W["xf"][cf], W["zf"][cf], -rtus.ALPHA_MAX, rtus.ALPHA_MAX)
            d = np.abs(got - ref)
            acc = {"max_abs_dt_s": float(np.nanmax(d)), "median_abs_dt_s": float(np.nanmedian(d)),
                   "nan_masks_identical": bool(np.array_equal(np.isnan(got), np.isnan(ref))),
                   "sample": "8 seeded-random elements x 2048 seeded-random targets", "checker": "orc_tt_lens (long double)",
                   "bar_s": 1e-9}
        return base, acc
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
                      f"oracle/rt_oracle.c orc_shoot_batch (fp64, O(N) polyline scan per ray, OpenMP)"}, None


def cpu_ref_path():
    """CPU legs of the reference-parity path on this host: the NumPy port (the reference's own O(N^2) scan as
    2-D array ops, 1 core) and the C port (OpenMP) — forward trace only, reference sweep geometry."""
    from oracle import cport, rt_numpy, rt_numpy_loop
    R = ref_inputs("ref_sweep")
    d = float(R["za"][0])
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 4.0:
        g = R["geoms"][k % len(R["geoms"])]
        rt_numpy.shoot(0.0, d, R["zf"], R["alpha"], g[0], g[1])
        k += 1
    np_rate = k * R["n"] / (time.perf_counter() - t0)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 6.0:                       # the reference's loop structure: a Python loop over the rays
        g = R["geoms"][k % len(R["geoms"])]
        rt_numpy_loop.shoot(0.0, d, R["zf"], R["alpha"], g[0], g[1])
        k += 1
    loop_rate = k * R["n"] / (time.perf_counter() - t0)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 4.0:
        cport.shoot_batch(R["xa"], R["za"], R["zf"], R["alpha"], R["geoms"])
        k += 1
    c_rate = k * R["geoms"].shape[0] * R["n"] / (time.perf_counter() - t0)
    return {"numpy_loop_port_rays_per_s": round(loop_rate, 1), "numpy_loop_port_cores": 1,
            "numpy_loop_port_is": "oracle/rt_numpy_loop.py: the reference's own structure — vectorised prologue / epilogue around a Python "
                                  "loop over the rays, ~10 NumPy calls on N-element arrays per ray (main_rt.py:384-393, 6-168), np.polyfit "
                                  "replaced by the two-point line (half of the reference's time is polyfit: SURVEY section 3.1)",
            "numpy_port_rays_per_s": round(np_rate, 1), "numpy_port_cores": 1,
            "numpy_port_is": "oracle/rt_numpy.py: the same O(N^2) scan as 2-D array operations (no per-ray Python loop)",
            "c_port_rays_per_s": round(c_rate, 1), "c_port_cores": cport.num_threads(),
            "note": "forward trace at N = 905 (reference measured in SURVEY: 5.3 k rays/s on one core)"}


# ------------------------------------------------------------------------------------ side measurements (N = 1)
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


def _event_ms(torch, fn, k, warm=3):
    """mean HIP-event time of k back-to-back calls on the current stream (ms)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


def extra_measurements(dev_api, dist_api, rtus, t64, torch, dev):
    """Side measurements (not the headline), one GPU: the other BASELINE configs and the reference-parity path."""
    import ctypes as C
    res = {}
    valu_doc, valu_src = load_profile_json("valu_{tag}.json")
    # --- the reference's own hot path: forward trace + matcher ------------------------------------------------
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
        rays = G * T * N
        entry = {"Mrays_per_s": round(rays / dt / 1e6, 2), "ms_per_pass": round(dt * 1e3, 4), "rays_per_pass": rays,
                 "what": "rtus_shoot_dev + rtus_match_dev (two calls, the per-ray arrays go through HBM), eager launches"}
        # the same work through the fused entry (rtus_sweep_dev: the matcher inside the trace kernel), with and without the per-ray arrays
        for tag, want in (("fused", ("tof", "land_x")), ("fused_hits_only", ())):
            fplan = dev_api.SweepPlan(G, T, N, R["x_rx"].size, want=want, params=rtus.Params(), fast=fast, atol=1e-6, rtol=1e-5)
            fpass = lambda: fplan.run(*a, x_rx)
            for _ in range(3):
                fpass()
            torch.cuda.synchronize()
            if tag == "fused":
                same = all(torch.equal(fplan.out[k].view(G * T, -1), box[0][i]) for i, k in enumerate(("first_ray", "hit", "tof_hit")))
            tg = Timed(torch, lambda s: fpass(), 20 if kind == "ref_sweep" else 5)
            dtg, msg = tg.run(lambda: None)
            entry[tag] = {"ms_per_pass_graph": round(msg, 5), "Mrays_per_s": round(rays / (msg * 1e-3) / 1e6, 2),
                          "ms_per_pass_eager": round(_best_ms(torch, fpass, 20 if kind == "ref_sweep" else 5), 4)}
            if tag == "fused":
                entry[tag]["equals_two_calls"] = bool(same)
            del fplan, tg
        tg = Timed(torch, lambda s: one_pass(), 20 if kind == "ref_sweep" else 5)
        entry["ms_per_pass_graph"] = round(tg.run(lambda: None)[1], 5)
        del tg
        if kind == "ref_scale":
            # roofline of the forward-trace kernel alone (HIP events around the trace, matcher excluded)
            ms = _event_ms(torch, lambda: plan.run(*a), 5)
            algb = rays * 16 + N * 16
            entry["roofline"] = {"bound": "hbm", "binds": "valu_issue (fp64 transcendentals + crossing search)",
                                 "kernel": f"rtus_shoot_kernel<{'true' if fast else 'false'}>", "avg_launch_ms": round(ms, 5),
                                 "algorithmic_bytes_per_launch": algb, "achieved": round(algb / (ms * 1e-3) / 1e9, 2),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                 "G_rays_per_s_kernel_only": round(rays / (ms * 1e-3) / 1e9, 2)}
            key = kind + ("_fastmath" if fast else "")
            if valu_doc and key in valu_doc:                              # what binds it: SQ counters of the same launch (profiles/)
                v = dict(valu_doc[key])
                v["source"] = valu_src + " (rocprofv3 --pmc SQ_* passes of scripts/run_shoot_once.py; not collected in this run)"
                t_issue = v["issue_cycles_per_pass"] / 1024.0 / (v["clock_ghz"] * 1e9)
                v["issue_bound_ms"] = round(t_issue * 1e3, 5)
                v["frac_of_issue_bound"] = round(t_issue * 1e3 / ms, 4)
                v["fp64_TFLOP_per_s"] = round(v["fp64_flop_per_pass"] / (ms * 1e-3) / 1e12, 2)
                v["frac_of_fp64_vector_peak"] = round(v["fp64_flop_per_pass"] / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, 4)
                entry["roofline_valu"] = v
            if not fast:
                # the drop-in's full result: all eight arrays of shoot_rays (main_rt.py:432-441) = 80 B per ray
                plan8 = dev_api.ShootPlan(G, T, N, want=("out8",), params=rtus.Params())
                ms8 = _event_ms(torch, lambda: plan8.run(*a), 5)
                alg8 = rays * 80 + N * 16
                entry["out8_variant"] = {"what": "the same trace storing all eight arrays of the reference's result dict (64 B written + alpha, "
                                                 "z_f read per ray)", "avg_launch_ms": round(ms8, 5), "G_rays_per_s_kernel_only": round(rays / (ms8 * 1e-3) / 1e9, 2),
                                         "roofline": {"bound": "hbm", "algorithmic_bytes_per_launch": alg8, "achieved": round(alg8 / (ms8 * 1e-3) / 1e9, 2),
                                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg8 / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
                del plan8
        res[kind + ("_fastmath" if fast else "")] = entry
    # --- BASELINE configs[2] again: the library's default accuracy tier, and the aperture handed over in random order ------
    W3 = planar_inputs("cfg3_planar", 0, 1)
    out3 = torch.empty((W3["n_e"], W3["n_f"]), dtype=torch.float64, device=dev)
    a3 = [t64(W3[k]) for k in ("xe", "ze", "xf", "zf")]
    n3 = W3["n_e"] * W3["n_f"]
    plan_acc = dev_api.LayersPlan(W3["z_if"], W3["c"], *a3, out=out3)
    for _ in range(3):
        plan_acc.run()
    torch.cuda.synchronize()
    timed = Timed(torch, lambda s: plan_acc.run(), 50)
    dta, msa = timed.run(lambda: None)
    res["cfg3_planar_accurate_tier"] = {"Mrays_per_s": round(n3 * 50 / dta / 1e6, 1), "ms_per_launch": round(msa, 5),
                                        "hbm_frac": round((n3 * 8 + 2 * (W3["n_e"] + W3["n_f"]) * 8) / (msa * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        "note": "rtus_tt_layers_dev, default tier (fp64 residual + Fermat expansion, <= 1e-13 relative)"}
    perm = np.random.default_rng(3).permutation(W3["n_e"])
    xsh, zsh = t64(W3["xe"][perm]), t64(W3["ze"][perm])
    ws = torch.empty(int(rtus.lib().rtus_tt_layers_sort_workspace_bytes(W3["n_e"])), dtype=torch.uint8, device=dev)
    shuffled = lambda taup: dev_api.tt_layers_sorted_dev(W3["z_if"], W3["c"], xsh, zsh, a3[2], a3[3], out=out3, ws=ws, taup=taup)
    plain_sh = lambda: dev_api.tt_layers_dev(W3["z_if"], W3["c"], xsh, zsh, a3[2], a3[3], out=out3)
    ent = {}
    for name, fn in (("sorted_entry_taup", lambda: shuffled(True)), ("sorted_entry_accurate", lambda: shuffled(False)), ("plain_entry_accurate", plain_sh)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        tm = Timed(torch, lambda s: fn(), 20)
        dts, mss = tm.run(lambda: None)
        ent[name] = {"Mrays_per_s": round(n3 * 20 / dts / 1e6, 1), "ms_per_launch": round(mss, 5)}
        del tm
    ent["note"] = ("the 256 elements in random order.  rtus_tt_layers_sorted_dev sorts the aperture on the device (one more launch, "
                   "O(n^2) rank) and stores each row where it belongs; plain_entry = rtus_tt_layers_dev on the same order: the predictor "
                   "has no neighbours to extrapolate from (cold-started Newton)")
    res["cfg3_planar_shuffled"] = ent
    del out3, plan_acc, timed
    # --- BASELINE configs[1] (the small planar launch: 17 MB of results, one 4-wave round per SIMD) ---------------
    W = planar_inputs("cfg2_planar", 0, 1)
    out2 = torch.empty((W["n_e"], W["n_f"]), dtype=torch.float64, device=dev)
    plan2 = dev_api.LayersPlan(W["z_if"], W["c"], t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"]), out=out2)
    for _ in range(5):
        plan2.run()
    torch.cuda.synchronize()
    timed = Timed(torch, lambda s: plan2.run(), 500)
    dt2, ms2 = timed.run(lambda: None)
    n2 = W["n_e"] * W["n_f"]
    res["cfg2_planar"] = {"Mrays_per_s": round(n2 * 500 / dt2 / 1e6, 1), "ms_per_launch": round(ms2, 5), "solves_per_launch": n2,
                          "hbm_frac": round(n2 * 8 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "500 back-to-back launches replayed as a hipGraph; the 16.8 MB result fits the 256 MiB Infinity "
                                  "Cache, so this 'HBM' fraction is fabric traffic, not DRAM traffic"}
    del out2, plan2, timed
    # the same problem size through the batched entry: 8 apertures (the array shifted in x) x one target grid per launch
    B = 8
    xe8 = t64(np.stack([W["xe"] + 0.3e-3 * b for b in range(B)]))
    ze8 = t64(np.zeros((B, W["n_e"])))
    xf8, zf8 = t64(W["xf"]), t64(W["zf"])
    out8 = torch.empty((B, W["n_e"], W["n_f"]), dtype=torch.float64, device=dev)
    runb = lambda: dev_api.tt_layers_batch_dev(W["z_if"], W["c"], xe8, ze8, xf8, zf8, out=out8)
    for _ in range(3):
        runb()
    torch.cuda.synchronize()
    timedb = Timed(torch, lambda s: runb(), 200)
    dtb, msb = timedb.run(lambda: None)
    res["cfg2_planar_batch8"] = {"Mrays_per_s": round(B * n2 * 200 / dtb / 1e6, 1), "ms_per_launch": round(msb, 5),
                                 "solves_per_launch": B * n2, "hbm_frac": round(B * n2 * 8 / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "rtus_tt_layers_batch_dev: 8 config-2-sized problems (8 apertures, one target grid) in ONE launch"}
    del out8, timedb
    # --- BASELINE configs[3]: curved-lens two-point Fermat solves, fp32: the whole 1024 x 1024^2 table on one GPU (the
    # N = 1 point of the strong-scaling series bench.py --gpus N runs) and the 128-row shard one of 8 GPUs solves ------
    lens = rtus.Params().lens()
    L = rtus.lib()
    for name, rows in (("cfg4_lens_f32_full", 1024), ("cfg4_lens_f32_shard128", 128)):
        Wl = lens_inputs(0, 1, n_rows=rows)
        mk = lambda a_: torch.as_tensor(np.ascontiguousarray(a_, dtype=np.float32), device=dev)
        txe, tze, txf, tzf = mk(Wl["xe"]), mk(Wl["ze"]), mk(Wl["xf"]), mk(Wl["zf"])
        outl = torch.empty((rows, Wl["n_f"]), dtype=torch.float32, device=dev)
        run = lambda: L.rtus_tt_lens_f32_dev(C.byref(lens), -rtus.ALPHA_MAX, rtus.ALPHA_MAX, txe.data_ptr(), tze.data_ptr(), rows,
                                             txf.data_ptr(), tzf.data_ptr(), Wl["n_f"], outl.data_ptr(), None,
                                             torch.cuda.current_stream().cuda_stream)
        assert run() == 0
        ms = _event_ms(torch, run, 10 if rows == 1024 else 40)
        n = rows * Wl["n_f"]
        res[name] = {"Mrays_per_s": round(n / ms / 1e3, 1), "ms_per_launch": round(ms, 4), "solves_per_launch": n,
                     "hbm_frac": round(n * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if rows == 1024:
            res.setdefault("_lens_samples", {})[name] = _table_sample(torch, outl, txe, tze, txf, tzf, seed=4)
        del outl
    # fp64 instantiation of the same kernel on a 1024 x (1024 x 256) strip
    Wl = lens_inputs(0, 1)
    mk = lambda a_: torch.as_tensor(np.ascontiguousarray(a_, dtype=np.float64), device=dev)
    nf64 = 1024 * 256
    txe, tze, txf, tzf = mk(Wl["xe"]), mk(Wl["ze"]), mk(Wl["xf"][:nf64]), mk(Wl["zf"][:nf64])
    outl = torch.empty((1024, nf64), dtype=torch.float64, device=dev)
    run = lambda: L.rtus_tt_lens_dev(C.byref(lens), -rtus.ALPHA_MAX, rtus.ALPHA_MAX, txe.data_ptr(), tze.data_ptr(), 1024,
                                     txf.data_ptr(), tzf.data_ptr(), nf64, outl.data_ptr(), None,
                                     torch.cuda.current_stream().cuda_stream)
    assert run() == 0
    ms = _event_ms(torch, run, 5)
    res["lens_fermat_f64"] = {"Mrays_per_s": round(1024 * nf64 / ms / 1e3, 1), "ms_per_launch": round(ms, 4),
                              "solves_per_launch": 1024 * nf64}
    res.setdefault("_lens_samples", {})["lens_fermat_f64"] = _table_sample(torch, outl, txe, tze, txf, tzf, seed=5)
    del outl
    # --- BASELINE configs[4]: the whole 2048 x 2048 FMC table on one GPU --------------------------------------------
    Wf = fmc_inputs(0, 1)
    outf = torch.empty((Wf["n_e"], Wf["n_f"]), dtype=torch.float64, device=dev)
    planf = dev_api.LayersPlan(Wf["z_if"], Wf["c"], t64(Wf["xe"]), t64(Wf["ze"]), t64(Wf["xf"]), t64(Wf["zf"]), out=outf)
    for _ in range(5):
        planf.run()
    torch.cuda.synchronize()
    timed = Timed(torch, lambda s: planf.run(), 200)
    dtf, msf = timed.run(lambda: None)
    nfm = Wf["n_e"] * Wf["n_f"]
    res["cfg5_fmc_full"] = {"Mrays_per_s": round(nfm * 200 / dtf / 1e6, 1), "ms_per_launch": round(msf, 5), "solves_per_launch": nfm,
                            "hbm_frac": round(nfm * 8 / (msf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del outf, planf, timed
    # --- consumers of a travel-time table: the memory-bound kernels of the library ------------------
    res.update(consumer_measurements(dev_api, t64, torch, dev))
    # --- root-finding pulse-echo solve (the north-star's replacement for the grid scan), device-resident ----------------
    res.update(solve_measurements(dev_api, rtus, t64, torch))
    # ... and once through the host-buffer API
    R = ref_inputs("ref_sweep")
    sol = [None]

    def one_solve():
        sol[0] = rtus.solve_travel_times(R["xa"], R["za"], R["x_rx"], R["alpha"], R["geoms"], params=rtus.Params())[0]

    dtm = _best_ms(torch, one_solve, 3, blocks=3) * 1e-3
    tt = sol[0]
    res["solve_sweep_host_api"] = {"ms_per_pass": round(dtm * 1e3, 3), "elements": int(tt.size), "M_element_solves_per_s": round(tt.size / dtm / 1e6, 1),
                                   "with_root": int(np.isfinite(tt).sum()), "note": "host-buffer API incl. PCIe"}
    # --- the NumPy-in / NumPy-out table call (PCIe inclusive; never `value`): configs[1] and configs[2], one device and two arenas ----
    for nm, wl_ in (("cfg2", "cfg2_planar"), ("cfg3", "cfg3_planar")):
        Wh = planar_inputs(wl_, 0, 1)
        outh = np.empty((Wh["n_e"], Wh["n_f"]))
        for label, kw in (("one_device", {}), ("devices_0_0", {"devices": [0, 0]}), ("one_device_taup", {"taup": True})):
            call = lambda: rtus.travel_time_layers(Wh["z_if"], Wh["c"], Wh["xe"], Wh["ze"], Wh["xf"], Wh["zf"], out=outh, **kw)
            ms_h = _best_ms(torch, call, 3, blocks=3)
            res.setdefault("host_api_tables", {})[f"{nm}_{label}"] = {
                "ms_per_call": round(ms_h, 3), "GB_per_s_of_results": round(outh.nbytes / (ms_h * 1e-3) / 1e9, 2),
                "Mrays_per_s": round(outh.size / ms_h / 1e3, 1)}
    res["host_api_tables"]["note"] = ("rtus.travel_time_layers: host buffers in, host table out — bound by the device-to-host copy of the table "
                                      "(pageable memory), not by the kernel; devices=[0, 0]: two arenas / streams / copy threads on the one GPU")
    # --- the reference's calling pattern: 210 sequential shoot_rays(N = 905) calls + matcher (main_rt.py:464-504) ------
    res["sweep_like_main_rt"] = sweep_like_reference(rtus)
    return res


def _table_sample(torch, table, txe, tze, txf, tzf, seed, n_rows=8, n_cols=2048):
    """a seeded sample of a device table with the coordinates the kernel saw (as float64) -> host arrays for the accuracy leg"""
    rng = np.random.default_rng(seed)
    re = np.sort(rng.choice(table.shape[0], size=n_rows, replace=False))
    cf = np.sort(rng.choice(table.shape[1], size=n_cols, replace=False))
    ri, ci = torch.as_tensor(re, device=table.device), torch.as_tensor(cf, device=table.device)
    f = lambda t, i: t[i].double().cpu().numpy()
    return dict(xe=f(txe, ri), ze=f(tze, ri), xf=f(txf, ci), zf=f(tzf, ci), got=table[ri][:, ci].double().cpu().numpy(),
                dtype=str(table.dtype).replace("torch.", ""))


def lens_tables_accuracy(samples):
    """(part of the CPU leg: the only place besides cpu_baseline / cpu_ref_path where bench.py touches oracle/) the curved-lens
    tables timed in `extra` — the whole fp32 table of configs[3] and the fp64 table — against orc_tt_lens on a seeded sample"""
    import rtus
    from oracle import cport
    out = {}
    for name, smp in samples.items():
        ref, _ = cport.tt_lens(smp["xe"], smp["ze"], smp["xf"], smp["zf"], -rtus.ALPHA_MAX, rtus.ALPHA_MAX)
        d = np.abs(smp["got"] - ref)
        out[name] = {"max_abs_dt_s": float(np.nanmax(d)), "median_abs_dt_s": float(np.nanmedian(d)), "dtype": smp["dtype"],
                     "nan_masks_identical": bool(np.array_equal(np.isnan(smp["got"]), np.isnan(ref))),
                     "sample": f"{smp['got'].shape[0]} seeded-random rows x {smp['got'].shape[1]} seeded-random targets of the timed table",
                     "checker": "orc_tt_lens (scan + golden section, long double)", "bar_s": 1e-9}
    return out


def solve_inputs(kind):
    """solve_sweep: the reference's own sweep (main_rt.py:464-482: 210 geometries, tx = the centre element, 65 rx, N = 905);
    solve_scale: the reference geometry scaled up — 16 pipe geometries x 1024 tx x 65 rx over the same N = 905 grid."""
    import rtus
    d = rtus.Params().d
    n = 905
    if kind == "solve_sweep":
        geoms = np.array([[r * 1e-2, o * 1e-3] for r in range(1, 11) for o in range(-10, 11)])
        xa = np.array([0.0])
    else:
        geoms = np.array([[0.02 + 0.005 * i, 0.0004 * (i - 7.5)] for i in range(16)])
        xa = (np.arange(1024) - 511.5) * 0.3e-4
    return dict(n=n, geoms=geoms, xa=xa, za=np.full(xa.size, d), alpha=np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n),
                x_rx=rtus.reference_elements())


def solve_measurements(dev_api, rtus, t64, torch):
    """rtus_solve_dev, inputs and outputs resident in HBM, K passes replayed as one hipGraph.  A pass = polyline + box
    records, grid trace (905 rays per (geometry, tx) row, with the bracket masks), bracket refinement: every element of every
    row gets its travel time(s).  `roofline` prices the pass against HBM as the contract asks (16 B written per element:
    travel time + launch angle); `roofline_valu` carries what binds it (SQ counters of the two kernels from profiles/)."""
    out = {}
    valu_doc, valu_src = load_profile_json("valu_{tag}.json")
    for kind, K in (("solve_sweep", 200), ("solve_scale", 10)):
        S = solve_inputs(kind)
        G, T, E, N = S["geoms"].shape[0], S["xa"].size, S["x_rx"].size, S["n"]
        a = [t64(S[k]) for k in ("geoms", "xa", "za", "alpha", "x_rx")]
        entry = {"elements_per_pass": G * T * E, "grid_rays_per_pass": G * T * N}
        for fast in (False, True):
            plan = dev_api.SolvePlan(G, T, N, E, params=rtus.Params(), fast=fast)
            for _ in range(3):
                o = plan.run(*a)
            torch.cuda.synchronize()
            nroot = int(torch.isfinite(o["tt"]).sum().item())
            timed = Timed(torch, lambda s: plan.run(*a), K)
            dt, ms = timed.run(lambda: None)
            timed_keep = Timed(torch, lambda s: plan.run(*a, polyline_ready=True), K)
            dtk, msk = timed_keep.run(lambda: None)
            el = G * T * E
            algb = el * 16 + (2 * G + 2 * T + N + E) * 8
            e2 = {"us_per_pass": round(ms * 1e3, 2), "M_element_solves_per_s": round(el / ms / 1e3, 1),
                  "us_per_pass_polyline_kept": round(msk * 1e3, 2), "M_element_solves_per_s_polyline_kept": round(el / msk / 1e3, 1),
                  "elements_with_a_ray_path": nroot,
                  "roofline": {"bound": "hbm", "binds": "launch floor + the fp64 dependent chain of a traced ray (few waves per SIMD)" if kind == "solve_sweep"
                               else "valu_issue (grid trace 2/3, refinement 1/3)",
                               "kernel": ("rtus_geom1_kernel + rtus_solve_row_kernel<%s> (one launch per pass after the polyline's: a workgroup per row, "
                                          "three lanes per bracket)" % ("true" if fast else "false")) if kind == "solve_sweep" else
                                         "rtus_shoot_kernel<%s, true> + rtus_solve_kernel<%s, true>" % (("true",) * 2 if fast else ("false",) * 2),
                               "algorithmic_bytes_per_pass": algb, "achieved": round(algb / (ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(algb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}}
            key = kind + ("_fastmath" if fast else "")
            if valu_doc and key in valu_doc:
                v = dict(valu_doc[key])
                v["source"] = valu_src + " (rocprofv3 --pmc SQ_* passes of scripts/run_solve_once.py; not collected in this run)"
                if v.get("issue_cycles_per_pass") and v.get("clock_ghz"):
                    t_issue = v["issue_cycles_per_pass"] / 1024.0 / (v["clock_ghz"] * 1e9)
                    v["issue_bound_us"] = round(t_issue * 1e6, 2)
                    v["frac_of_issue_bound"] = round(t_issue * 1e3 / ms, 4)
                if v.get("fp64_flop_per_pass"):
                    v["fp64_TFLOP_per_s"] = round(v["fp64_flop_per_pass"] / (ms * 1e-3) / 1e12, 2)
                    v["frac_of_fp64_vector_peak"] = round(v["fp64_flop_per_pass"] / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, 4)
                e2["roofline_valu"] = v
            if kind == "solve_sweep":                      # the same pass as separate launches and with the one-lane iteration (round 3's path)
                for label, kw in (("three_launches", {"three_launches": True}), ("one_lane_three_launches", {"one_lane": True})):
                    p2 = dev_api.SolvePlan(G, T, N, E, params=rtus.Params(), fast=fast, **kw)
                    for _ in range(3):
                        p2.run(*a)
                    torch.cuda.synchronize()
                    t2 = Timed(torch, lambda s: p2.run(*a), K)
                    _, ms2 = t2.run(lambda: None)
                    e2["us_per_pass_" + label] = round(ms2 * 1e3, 2)
                    del p2, t2
            entry["vector_form" if fast else "reference_arithmetic"] = e2
            del plan, timed, timed_keep
        entry["note"] = ("polyline_kept: RTUS_POLYLINE_READY — the lens polyline of the alpha grid stays in the workspace between passes "
                         "(main_rt.py recomputes the same x_p, z_p for each of its 210 geometries)")
        out[kind] = entry
    return out


def consumer_measurements(dev_api, t64, torch, dev):
    """Focal laws over the configs[2] table (HBM-bound stream) and a TFM delay-and-sum (64 x 64 A-scans of 2048 samples,
    256 x 256 image; L2-gather-bound), each with its own roofline."""
    res = {}
    W = planar_inputs("cfg3_planar", 0, 1)
    tt3 = dev_api.tt_layers_dev(W["z_if"], W["c"], t64(W["xe"]), t64(W["ze"]), t64(W["xf"]), t64(W["zf"]))
    out = torch.empty_like(tt3)
    ms = _event_ms(torch, lambda: dev_api.focal_delays_dev(tt3, out=out), 10)
    n = tt3.numel()
    res["focal_delays_cfg3"] = {"ms_per_launch": round(ms, 4), "entries": n,
                                "roofline": {"bound": "hbm", "kernel": "rtus_focal_delays_once_kernel<32>", "algorithmic_bytes_per_launch": 16 * n,
                                             "achieved": round(16 * n / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": round(16 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                             "note": "8 B read + 8 B written per entry: the column stays in registers between the maximum and "
                                                     "the subtraction (apertures up to 512 elements)"}}
    del tt3, out
    n_el, n_t, fs = 64, 2048, 50e6
    xe = (np.arange(n_el) - (n_el - 1) / 2.0) * 0.6e-3
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 256), np.linspace(0.025, 0.045, 256))
    tt = dev_api.tt_layers_dev([0.020], [2330.0, 1483.0], t64(xe), t64(np.zeros(n_el)), t64(xs.ravel()), t64(zs.ravel()))
    fmc = torch.randn((n_el, n_el, n_t), dtype=torch.float32, device=dev)
    img = torch.empty(xs.size, dtype=torch.float32, device=dev)
    ms = _event_ms(torch, lambda: dev_api.tfm_dev(fmc, fs, tt, out=img), 10)
    pairs = n_el * n_el * xs.size
    res["tfm_64x64x2048_256x256"] = {
        "ms_per_launch": round(ms, 4), "pair_samples_per_launch": pairs, "G_pair_samples_per_s": round(pairs / (ms * 1e-3) / 1e9, 1),
        "roofline": {"bound": "l2-gather", "kernel": "rtus_tfm_kernel", "gathered_bytes_per_launch": 8 * pairs,
                     "achieved": round(8 * pairs / (ms * 1e-3) / 1e9, 1), "peak": 16800.0, "unit": "GB/s (gathers served by the XCDs' L2s)",
                     "frac": round(8 * pairs / (ms * 1e-3) / 1e9 / 16800.0, 4),
                     "hbm_algorithmic_bytes_per_launch": fmc.numel() * 4 + tt.numel() * 8 + img.numel() * 4,
                     "traffic": (load_profile_json("traffic_{tag}.json")[0] or {}).get("tfm_bytes_per_launch"),
                     "note": "two neighbouring fp32 samples (one 8-byte load) per (tx, rx, focal point).  peak: MI355X_MICROARCH.md, 'Indexed "
                             "rows: gather' — rows served from the XCD's L2 16.8-18.8 TB/s chip-wide (lower end taken); a 38 MB table gathered "
                             "at random (Infinity Cache) reaches 8.6 TB/s.  The 34 MB FMC block does not fit an L2 (4 MiB), but with XCD k on a "
                             "contiguous eighth of the focal points each L2 only sees its eighth of the sample windows: L2-miss traffic "
                             "(`traffic`, FETCH_SIZE x 2 + WRITE_SIZE) is ~1x the algorithmic bytes (round 2: 2.5x).  What binds the kernel is "
                             "the vector-memory address path (64 scattered 8-byte requests per wave-instruction), not bandwidth"}}
    return res


def sweep_like_reference(rtus):
    """The reference's own driver loop, call for call (main_rt.py:464-504): 210 x shoot_rays(x_a[32], z_a[32], zf, alpha)
    through the host-buffer API (NumPy in, dict of 8 NumPy arrays out) + the 65-element matcher per geometry.
    The reference takes 141 s for this on one Xeon core."""
    d = rtus.Params().d
    n = 905
    x_a = rtus.reference_elements()
    alpha = np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)
    zf = np.full(n, d)
    geoms = [(r * 1e-2, o * 1e-3) for r in range(1, 11) for o in range(-10, 11)]
    rtus.shoot_rays(x_a[32], d, zf, alpha, params=rtus.Params(r_outer=0.05, pipe_offset=0.0))      # warm
    t0 = time.perf_counter()
    hits = 0
    for r_o, off in geoms:
        p = rtus.Params(r_outer=r_o, pipe_offset=off)
        res = rtus.shoot_rays(x_a[32], d, zf, alpha, plot=False, params=p)
        b = rtus.shoot_batch([x_a[32]], [d], zf, alpha, params=p, want=("tof", "land_x"))
        hit, tof, _ = rtus.match_elements(b["land_x"][0, 0], b["tof"][0, 0], x_a, atol=1e-6)
        hits += int(hit.sum())
    wall = time.perf_counter() - t0
    t1 = time.perf_counter()
    for r_o, off in geoms[:50]:
        rtus.shoot_rays(x_a[32], d, zf, alpha, plot=False, params=rtus.Params(r_outer=r_o, pipe_offset=off))
    per_call = (time.perf_counter() - t1) / 50
    return {"wall_s": round(wall, 4), "calls": len(geoms), "hits": hits, "shoot_rays_us_per_call": round(per_call * 1e6, 1),
            "reference_wall_s": 141.0, "note": "210 sequential shoot_rays + shoot_batch + match_elements calls through the "
            "host-buffer API (PCIe + launch latency per call); reference: 141 s on one Xeon core (SURVEY section 6)"}


if __name__ == "__main__":
    main()

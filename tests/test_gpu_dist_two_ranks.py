"""GPU: the sharded travel-time tables with the REAL HIP solvers across two ranks.  This pool gives one GPU per
box, so both ranks share cuda:0 and exchange over gloo (RCCL needs one device per rank); what is checked is the
product's sharding + gather code path end to end: the reassembled matrix is BYTE-IDENTICAL to the one-process result
(shards start on the table's workgroup-block boundaries and the kernels are told where their rows sit in the table, so a
row is solved in the same workgroup with the same predecessors either way), for the planar-layer solver (fp64) and the
curved-lens solver (fp32), with row counts that need padding."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import D_PLANE, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from importlib import import_module
    import rtus
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = import_module("ray-tracing-ultrasound_amd.dist")
    t = lambda a, dt=np.float64: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device="cuda")
    n_e = 37                                                   # not divisible by 2 -> pad rows
    xe = (np.arange(n_e) - (n_e - 1) / 2) * 0.6e-3
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 40), np.linspace(0.025, 0.065, 30))
    full = d.travel_time_layers_sharded([0.02], [2330.0, 1483.0], t(xe), t(np.zeros(n_e)), t(xs.ravel()), t(zs.ravel()))
    # a table big enough for multi-row workgroups (8 rows per block here): 150 rows -> shards of 80 + 70
    n_b = 150
    xb = (np.arange(n_b) - (n_b - 1) / 2) * 0.25e-3
    xg, zg = np.meshgrid(np.linspace(-0.02, 0.02, 160), np.linspace(0.026, 0.066, 128))
    big = d.travel_time_layers_sharded([0.010, 0.025], [2330.0, 1483.0, 5900.0], t(xb), t(np.zeros(n_b)), t(xg.ravel()), t(zg.ravel()))
    xl, zl = np.meshgrid(np.linspace(-0.004, 0.004, 33), np.linspace(0.03, 0.07, 21))
    lens = d.travel_time_lens_sharded(t(xe * 0.1, np.float32), t(np.full(n_e, D_PLANE), np.float32),
                                      t(xl.ravel(), np.float32), t(zl.ravel(), np.float32), params=rtus.Params())
    torch.cuda.synchronize()
    xl2, zl2 = np.meshgrid(np.linspace(-0.004, 0.004, 256), np.linspace(0.03, 0.07, 128))
    lens_big = d.travel_time_lens_sharded(t(xb * 0.1, np.float32), t(np.full(n_b, D_PLANE), np.float32),
                                          t(xl2.ravel(), np.float32), t(zl2.ravel(), np.float32), params=rtus.Params())
    torch.cuda.synchronize()
    if rank == 0:
        q.put((full.cpu().numpy(), lens.cpu().numpy(), big.cpu().numpy(), lens_big.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharded_tables_equal_single_process(rtus):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, lens, big, lens_big = q.get()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    n_e = 37
    xe = (np.arange(n_e) - (n_e - 1) / 2) * 0.6e-3
    xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 40), np.linspace(0.025, 0.065, 30))
    one = rtus.travel_time_layers([0.02], [2330.0, 1483.0], xe, np.zeros(n_e), xs.ravel(), zs.ravel())
    assert full.shape == one.shape
    assert np.array_equal(full, one)                       # byte-identical (SURVEY section 7 test plan)
    xl, zl = np.meshgrid(np.linspace(-0.004, 0.004, 33), np.linspace(0.03, 0.07, 21))
    one32 = rtus.travel_time_lens(xe * 0.1, np.full(n_e, D_PLANE), xl.ravel(), zl.ravel(), params=rtus.Params(),
                                  dtype=np.float32)
    assert lens.shape == one32.shape and lens.dtype == np.float32
    assert np.array_equal(lens, one32)
    # multi-row workgroups: the shard boundary (row 80) is a block boundary of the whole table
    n_b = 150
    xb = (np.arange(n_b) - (n_b - 1) / 2) * 0.25e-3
    xg, zg = np.meshgrid(np.linspace(-0.02, 0.02, 160), np.linspace(0.026, 0.066, 128))
    from importlib import import_module
    import torch
    dev = import_module("ray-tracing-ultrasound_amd.device")
    assert dev.rows_per_block(n_b, xg.size) > 1
    one_b = rtus.travel_time_layers([0.010, 0.025], [2330.0, 1483.0, 5900.0], xb, np.zeros(n_b), xg.ravel(), zg.ravel())
    assert np.array_equal(big, one_b)
    xl2, zl2 = np.meshgrid(np.linspace(-0.004, 0.004, 256), np.linspace(0.03, 0.07, 128))
    assert dev.rows_per_block(n_b, xl2.size, torch.float32) > 1
    one_l = rtus.travel_time_lens(xb * 0.1, np.full(n_b, D_PLANE), xl2.ravel(), zl2.ravel(), params=rtus.Params(), dtype=np.float32)
    assert np.array_equal(lens_big, one_l)


def _run_bench_two_ranks(extra_args):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank, started BEFORE anything
    touches the GPU), with the one-GPU rehearsal switches: both ranks on cuda:0, gloo instead of RCCL, small tables."""
    import json
    import subprocess
    env = dict(os.environ, RTUS_BENCH_ONE_GPU="1", RTUS_BENCH_BACKEND="gloo", RTUS_BENCH_SMALL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"] + extra_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_default_multi_gpu_line_strong_scaled_with_reassembly():
    """The N > 1 control flow of bench.py: BASELINE configs[3] strong-scaled (tx rows / N), collective-free timed region,
    the reassembly all-gather timed on its own and folded into value_with_reassembly, configs[4] under extra."""
    d = _run_bench_two_ranks([])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dtype"] == "f32" and d["steps"] == 4
    assert "configs[3]" in d["config"]["workload"] and d["config"]["solves_per_step_all_gpus"] == 2 * d["config"]["solves_per_step_per_gpu"]
    assert d["value"] > 0 and d["reassembly"]["ms"] > 0
    assert 0 < d["value_reassembled_every_step"] < d["value_with_reassembly"] < d["value"]
    assert d["roofline"]["kernel"].startswith("rtus_tt_lens_kernel") and 0 < d["roofline"]["frac"] < 1
    f = d["extra"]["cfg5_fmc"]
    assert "error" not in f and f["rows_per_gpu"] * 2 >= 64 and f["reassembly_ms"] > 0


def test_bench_weak_scaled_planar_and_gather_modes():
    d = _run_bench_two_ranks(["--workload", "cfg3_planar", "--gather", "end", "--no-extra"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "configs[2]" in d["config"]["workload"]
    assert "inside the timed region" in d["config"]["sharding"] and "reassembly" not in d
    d = _run_bench_two_ranks(["--workload", "cfg3_planar", "--gather", "step", "--no-extra"])
    assert "every step" in d["config"]["sharding"] and d["config"]["launch"] == "eager launches"


def test_bench_reference_sweep_workload_two_ranks():
    """--workload ref_sweep at N = 2: every rank runs the fused trace + matcher entry on its own copy of the sweep (weak scaling, no
    collective), rank 0 prints the line."""
    d = _run_bench_two_ranks(["--workload", "ref_sweep", "--no-extra", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "true>" in d["roofline"]["kernel"] and "reference sweep" in d["config"]["workload"] and "no exchange" in d["config"]["sharding"]


def _tfm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from importlib import import_module
    from oracle import tfm_numpy
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = import_module("ray-tracing-ultrasound_amd.dist")
    n_el, n_t, fs = 16, 1500, 40e6
    x = (np.arange(n_el) - 7.5) * 0.6e-3
    fmc = tfm_numpy.synth_fmc(x, np.zeros(n_el), [(0.002, 0.018, 1.0)], 1483.0, fs, n_t)
    xs, zs = np.meshgrid(np.linspace(-0.006, 0.006, 67), np.linspace(0.012, 0.024, 45))            # 3015 focal points
    t = lambda a, dt=np.float64: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device="cuda")
    img = d.tfm_layers_sharded(t(fmc, np.float32), fs, [0.004], [2330.0, 1483.0], t(x), t(np.zeros(n_el)), t(xs.ravel()), t(zs.ravel()))
    torch.cuda.synchronize()
    if rank == 0:
        q.put(img.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_column_sharded_tfm_equals_single_process(rtus):
    """The consumer path that needs no table reassembly (dist.tfm_layers_sharded): focal points split over two ranks,
    HIP Fermat + TFM kernels per slice, image slices gathered == the one-process image."""
    import torch.multiprocessing as mp
    from oracle import tfm_numpy
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_tfm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    img = q.get()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    n_el, n_t, fs = 16, 1500, 40e6
    x = (np.arange(n_el) - 7.5) * 0.6e-3
    fmc = tfm_numpy.synth_fmc(x, np.zeros(n_el), [(0.002, 0.018, 1.0)], 1483.0, fs, n_t)
    xs, zs = np.meshgrid(np.linspace(-0.006, 0.006, 67), np.linspace(0.012, 0.024, 45))
    tt = rtus.travel_time_layers([0.004], [2330.0, 1483.0], x, np.zeros(n_el), xs.ravel(), zs.ravel())
    one = rtus.tfm_image(fmc, fs, tt)
    assert img.shape == one.shape
    assert np.max(np.abs(img - one)) <= 1e-5 * np.max(np.abs(one))          # (slices start Newton from other predictors)

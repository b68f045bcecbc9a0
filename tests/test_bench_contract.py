"""CPU: the parts of bench.py that do not need a GPU — argument defaults the driver relies on, the synthetic inputs of
every BASELINE config (shapes, weak-scaling shards), and the refusal to run without a GPU (no CPU fallback)."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_and_flags(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    # N = 1 headline: BASELINE configs[2], the largest single-GPU configuration
    assert a.gpus == 1 and a.steps == 500 and a.warmup == 20 and a.workload == "cfg3_planar"
    assert a.gather == "after" and a.graph == "on" and a.streams == 1
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "3"])
    a = bench.parse()
    # N > 1 headline: BASELINE configs[3] strong-scaled
    assert (a.gpus, a.steps, a.warmup, a.workload) == (8, 7, 3, "cfg4_lens_f32")
    assert bench.scaling_of("cfg4_lens_f32", 8) == bench.scaling_of("cfg5_fmc", 8) == "strong"
    assert bench.scaling_of("cfg3_planar", 8) == "weak"
    a = bench.parse(["--gpus", "4", "--workload", "cfg3_planar"])
    assert a.workload == "cfg3_planar"


def test_config_inputs_shapes_and_shards():
    w = bench.planar_inputs("cfg2_planar", 0, 1)
    assert w["n_e"] == 128 and w["n_f"] == 128 * 128 and len(w["c"]) == len(w["z_if"]) + 1 == 2
    assert w["xe"].shape == (128,) and w["xf"].shape == w["zf"].shape == (16384,)
    w3 = bench.planar_inputs("cfg3_planar", 0, 1)
    assert w3["n_e"] == 256 and w3["n_f"] == 512 * 512 and len(w3["c"]) == 3
    # weak scaling: rank r owns block r of a world*n_e aperture, same focal grid on every rank
    full = np.concatenate([bench.planar_inputs("cfg2_planar", r, 8)["xe"] for r in range(8)])
    assert full.size == 1024 and np.allclose(np.diff(full), 0.6e-3) and abs(full.mean()) < 1e-12
    assert np.array_equal(bench.planar_inputs("cfg2_planar", 3, 8)["xf"], w["xf"])
    # strong scaling: the 2048-row FMC table and the 1024-row lens table are fixed, rows are split over the ranks
    f = bench.fmc_inputs(1, 8)
    assert f["n_e"] == 256 and f["lo"] == 256 and f["rows_total"] == 2048 and f["n_f"] == 2048
    assert len(f["c"]) == len(f["z_if"]) + 1 == 5 and np.all(np.diff(f["z_if"]) > 0)
    assert bench.fmc_inputs(0, 1)["n_e"] == 2048
    rows = np.concatenate([bench.fmc_inputs(r, 3)["xe"] for r in range(3)])          # 2048 = 683 + 683 + 682
    assert np.array_equal(rows, bench.fmc_inputs(0, 1)["xe"])
    r = bench.ref_inputs("ref_sweep")
    assert r["geoms"].shape == (210, 2) and r["n"] == 905 and r["x_rx"].size == 65
    r = bench.ref_inputs("ref_scale")
    assert r["xa"].size == 1024 and r["n"] == 8192


def test_refuses_to_run_without_a_gpu():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    if p.returncode == 0:            # a GPU box that ignores the masking variables: nothing to check here
        return
    assert "needs a GPU" in (p.stderr + p.stdout)

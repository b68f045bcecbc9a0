"""GPU: the crossing search's hand-written first pass (GCN assembly on fixed scalar registers, csrc/rtus_trace.h
walk_first_pass) against its C++ twin: the same library built with -DRTUS_WALK_CXX (variants/librtus_walkcxx.so, built by
__graft_entry__.build()) must produce the SAME BITS on the reference goldens' inputs and on long / ragged polylines.  A compiler
update that breaks the assembly's register assumptions cannot pass silently."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import D_PLANE, ROOT, load_golden

pytestmark = pytest.mark.gpu

CHILD = r"""
import sys, os, numpy as np
sys.path.insert(0, %r)
import rtus
D = %r
out = {}
g = np.load(os.path.join(%r, "tests", "golden", "sweep_cfg.npz"))
geoms = g["geoms"][::7]
b = rtus.shoot_batch([0.0, -0.0123], [D, D], np.full(905, D), g["alpha"], geoms, params=rtus.Params(), want=("out8", "tof"))
out["sweep8"], out["sweep_tof"] = b["out8"], b["tof"]
rng = np.random.default_rng(9)
for n in (2, 9, 64, 513, 4097, 20011):
    al = np.sort(rng.uniform(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)) if n %% 2 else np.linspace(-rtus.ALPHA_MAX, rtus.ALPHA_MAX, n)
    out["n%%d" %% n] = rtus.shoot_batch([0.003], [D], np.full(n, D), al, [[0.037, 0.0038], [0.02, 0.0], [0.01, -0.01]], params=rtus.Params())["out8"]
tt = rtus.solve_travel_times([0.0], [D], g["x_elem"], g["alpha"], geoms, params=rtus.Params(), all_roots=True)
out["solve_tt"], out["solve_all"] = tt[0], tt[2]
np.savez(sys.argv[1], **out)
""" % (ROOT, D_PLANE, ROOT)


def _run(lib, path):
    env = dict(os.environ)
    env.pop("RTUS_LIB", None)
    if lib:
        env["RTUS_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD, path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return np.load(path)


def test_assembly_walk_and_cxx_walk_give_the_same_bits(tmp_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    variant = entry.build_walk_cxx_variant()                         # rebuilt here if missing or made from other sources than this tree's
    a = _run(None, str(tmp_path / "asm.npz"))
    b = _run(variant, str(tmp_path / "cxx.npz"))
    assert set(a.files) == set(b.files) and len(a.files) >= 9
    for k in a.files:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.isfinite(a["sweep_tof"]).sum() > 1000

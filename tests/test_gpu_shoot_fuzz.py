"""GPU: randomized comparisons against the C oracle.  Forward trace (scripts/fuzz_shoot.py, 40 trials here; long runs
are quoted in DESIGN.md): all hierarchy depths of the crossing search, tangent pipes, zero offset, off-centre elements,
random launch grids.  Hard criterion per ray and output: |gpu - oracle| <= 1e-12 m + 16 x the oracle's own spread under
1-2 ulp nudges of its inputs; NaN masks identical on every ray whose NaN status the oracle keeps under those nudges.  The
self-test run moves every 97th ray onto the next polyline chord (off-by-one segment index) and must be caught."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_forward_trace_vs_oracle(rtus):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_shoot.py"), "40", "777"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK: 40 trials" in r.stdout
    # the first rule (fourteen noise patterns) is the gate: a trace that needs the second look is counted, not waved through
    import re
    m = re.search(r"SECOND LOOKS: (\d+) trace\(s\) of (\d+) failed the first rule .*?, (\d+) of them passed", r.stdout)
    assert m, r.stdout[-500:]
    print(r.stdout.strip().splitlines()[-1])
    assert int(m.group(1)) == 0, f"{m.group(1)} of {m.group(2)} traces failed the first rule (second look passed {m.group(3)})"


def test_fuzz_criterion_catches_a_wrong_segment(rtus):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_shoot.py"), "6", "4242", "--selftest"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, r.stdout[-2000:] + r.stderr[-2000:]          # 3 = the corruption was caught
    assert "MISMATCH" in r.stdout


def test_fuzz_planar_layers_vs_oracle(rtus):
    """scripts/fuzz_layers.py: random media (0-8 interfaces), apertures (shuffled / duplicated / clustered positions,
    changing depths, elements inside layers), targets and sizes; 60 trials here (long run: 400 trials, 10.7 M solves,
    worst 4.7e-17 s / 1.9e-12 relative).  NaN masks identical, |dt| <= 1e-16 s + 2e-11 t."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_layers.py"), "60", "31"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK: 60 trials" in r.stdout


def test_fuzz_planar_layers_taup_tier_vs_oracle(rtus):
    """The same fuzz through the tau-p tier (device-side sorted entry), bar 1e-16 s + 6e-11 t; every fourth trial is a table large
    enough for >= 8 rows per workgroup (asserted in the script): the four-history runs and the held groups of rtus_fermat.hip.
    (Those trials were added in round 4 and found round 3's tier 2.8e-10 off on a coarse random aperture — the reciprocal of X' taken
    by one Newton step from the previous element's — and the first version of the held groups 5.8e-10 off under a target.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_layers.py"), "48", "77", "--taup"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK (tau-p tier, sorted entry): 48 trials" in r.stdout


def test_fuzz_lens_kernels_vs_oracle(rtus):
    """scripts/fuzz_lens.py: curved-lens Fermat kernels (fp64 + fp32) vs the golden-section oracle on random apertures /
    targets / sizes; 25 trials here (long run: 150 trials, 0.2 M solves, worst 4.7e-20 s fp64, 2.1e-11 s fp32)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_lens.py"), "25", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK: 25 trials" in r.stdout


def test_fuzz_root_finding_solve_vs_oracle(rtus):
    """scripts/fuzz_solve.py: rtus_solve in its three forms (one launch / three launches with three lanes per bracket, one lane per
    bracket) against the oracle's bisection on random geometries (centred and near-tangent pipes too), transmit points, apertures and
    grids; 45 trials here (long run: profiles/README.md).  Root counts equal except where a branch ends inside a bracket (<= 1 % of
    the elements), roots within 1e-13 s / 1e-11 rad.  (It found round 4's first three-lane acceptance rule reporting a root at every
    JUMP of x_land — a sign change across a discontinuity — for whole rows of near-tangent geometries: the triple must be continuous.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_solve.py"), "45", "11"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK: 45 trials" in r.stdout


def test_planar_exact_boundary_cases_vs_oracle(rtus):
    """scripts/exp_planar_exact_cases.py: what a random fuzz never draws — targets and elements EXACTLY on interface depths, targets
    exactly below elements (dx = 0) and at the elements' own x, a whole table above its aperture, half of one — on tables with >= 8 rows
    per workgroup, both tiers, the plain and the sorted entry: identical NaN masks, |dt| <= 2e-11 t (tau-p tier: 6e-11 t)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exp_planar_exact_cases.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAILURES: 0" in r.stdout


def test_lens_exact_symmetry_cases_vs_oracle(rtus):
    """scripts/exp_lens_exact_cases.py: the curved-lens tables on a mirror-symmetric aperture and grid with an on-axis element and an
    on-axis target column (x = 0 exactly), target columns at the elements' own x, every position three times, and a symmetric window
    through the focus — fp64 <= 1e-15 s, fp32 <= 2e-10 s against the global-minimum oracle, >= 8 rows per workgroup."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exp_lens_exact_cases.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAILURES: 0" in r.stdout

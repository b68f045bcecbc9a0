"""GPU: the curved-lens travel-time TABLES on the code path BASELINE configs[3] runs — many rows per workgroup (continuation from
element to element, rows that take T alone at an extrapolated start), no alpha output, fp64 and fp32 — against the oracle on
WHOLE ROWS, including targets around the lens focus, where T(alpha) is nearly flat (the lens is aplanatic) and an off-axis
element has two local minima: the table entry is the LEAST time (Fermat), whichever minimum a continuation would follow.

Geometry: the reference's lens h(alpha), main_rt.py:180-234.  Checker: oracle.cport.tt_lens (scan + golden-section search in long
double; nothing shared with the kernel's Newton iteration).  Tolerances, on every entry, none excluded: fp64 |dt| < 1e-15 s,
fp32 |dt| < 2e-10 s (north-star: 1e-9 s).
"""
import numpy as np
import pytest

from conftest import D_PLANE

pytestmark = pytest.mark.gpu

# the three focus windows x three apertures of scripts/exp_lens_focus_accuracy.py (round 3, scratch) promoted to a test
WINDOWS = ((0.0, 0.0, 2e-3), (0.0, 0.004, 4e-3), (0.003, 0.001, 1e-3))
APERTURES = ((1024, 3e-5, 256), (256, 1.2e-4, 512), (128, 3e-4, 725))   # elements, pitch, targets per side: >= 64 rows per workgroup


def _f32_exact(v):
    """coordinates both precisions (and the oracle) can hold exactly"""
    return np.asarray(v, dtype=np.float32).astype(np.float64)


def _tables(dev, torch, xe, ze, xf, zf):
    import rtus
    PARAMS = rtus.Params()
    out = {}
    for dt in (torch.float64, torch.float32):
        t = [torch.tensor(v, dtype=dt, device="cuda") for v in (xe, ze, xf, zf)]
        tab = torch.empty((xe.size, xf.size), dtype=dt, device="cuda")
        dev.tt_lens_rows_dev(*t, tab, params=PARAMS)        # no alpha output: the T-only rows
        torch.cuda.synchronize()
        out[dt] = tab
    return out


def _rows_to_check(n_e, eb, n_f, budget=1.3e6):
    cand = [0, 1, 2, 3, 4, 5, 6, 7, 8, eb // 2, eb - 2, eb - 1, eb, eb + 4, n_e - 1, n_e // 2 + 5]
    cand = sorted({r for r in cand if 0 <= r < n_e})
    keep = max(4, int(budget // n_f))
    if len(cand) > keep:                                     # spread over the candidates, first and last kept
        idx = np.unique(np.round(np.linspace(0, len(cand) - 1, keep)).astype(int))
        cand = [cand[i] for i in idx]
    return cand


def _check(rtus, dev, torch, xe, ze, xf, zf, min_rows, label):
    from oracle import cport
    eb64 = dev.rows_per_block(xe.size, xf.size)
    eb32 = dev.rows_per_block(xe.size, xf.size, torch.float32)
    assert eb64 >= min_rows and eb32 >= min_rows, (eb64, eb32)      # the multi-row path, not one row per workgroup
    tabs = _tables(dev, torch, xe, ze, xf, zf)
    rows = _rows_to_check(xe.size, eb64, xf.size)
    ref, aref = cport.tt_lens(xe[rows], ze[rows], xf, zf, -rtus.ALPHA_MAX, rtus.ALPHA_MAX)
    pinned = np.abs(aref) > rtus.ALPHA_MAX - 1e-9
    ridx = torch.tensor(rows, device="cuda")
    t64 = tabs[torch.float64][ridx].cpu().numpy()
    t32 = tabs[torch.float32][ridx].cpu().numpy().astype(np.float64)
    d64, d32 = np.abs(t64 - ref), np.abs(t32 - ref)
    print(f"{label}: rows/workgroup {eb64}/{eb32}, rows {rows}, {ref.size} entries ({pinned.mean():.1%} at an end of the interval): "
          f"fp64 max {d64.max():.2e} s, fp32 max {d32.max():.2e} s mean {d32.mean():.2e} s")
    assert np.isfinite(t64).all() and np.isfinite(t32).all()
    assert d64.max() < 1e-15, (label, "fp64", d64.max(), np.unravel_index(d64.argmax(), d64.shape))
    assert d32.max() < 2e-10, (label, "fp32", d32.max(), np.unravel_index(d32.argmax(), d32.shape))
    return pinned


@pytest.mark.parametrize("n_e,pitch,g", APERTURES)
def test_focus_windows_whole_rows_vs_oracle(rtus, n_e, pitch, g):
    import torch
    from rtus import device as dev
    xe = _f32_exact((np.arange(n_e) - (n_e - 1) / 2) * pitch)
    ze = _f32_exact(np.full(n_e, D_PLANE))
    some_pinned = some_interior = False
    for x0, z0, half in WINDOWS:
        xl, zl = np.meshgrid(np.linspace(x0 - half, x0 + half, g), np.linspace(max(z0 - half, 1e-5), z0 + half, g))
        pinned = _check(rtus, dev, torch, xe, ze, _f32_exact(xl.ravel()), _f32_exact(zl.ravel()), 64,
                        f"n_e {n_e} pitch {pitch:g} window ({x0:g}, {z0:g}) +- {half:g}")
        some_pinned |= bool(pinned.any())
        some_interior |= bool((~pinned).any())
    assert some_pinned and some_interior                     # both kinds of minimum were in the comparison


def test_config4_like_table_whole_rows_vs_oracle(rtus):
    """The insonified region BASELINE configs[3] tabulates (x within +-4 mm, z 30-70 mm), 1024 elements @ 0.03 mm, enough targets
    for 128 rows per workgroup in fp32."""
    import torch
    from rtus import device as dev
    xe = _f32_exact((np.arange(1024) - 511.5) * 0.3e-4)
    ze = _f32_exact(np.full(1024, D_PLANE))
    xs, zs = np.meshgrid(np.linspace(-0.004, 0.004, 363), np.linspace(0.03, 0.07, 363))
    xf, zf = _f32_exact(xs.ravel()), _f32_exact(zs.ravel())
    assert dev.rows_per_block(1024, xf.size, torch.float32) == 128
    _check(rtus, dev, torch, xe, ze, xf, zf, 64, "configs[3]-like")


def test_beyond_the_focus_both_ends_compete(rtus):
    """Targets BEYOND the focus (z < 0): T(alpha) has its maximum in the middle and a minimum at each end of the interval; the
    lesser end is the entry."""
    import torch
    from rtus import device as dev
    xe = _f32_exact((np.arange(1024) - 511.5) * 0.3e-4)
    ze = _f32_exact(np.full(1024, D_PLANE))
    xs, zs = np.meshgrid(np.linspace(-0.003, 0.003, 256), np.linspace(-0.012, -0.0005, 256))
    pinned = _check(rtus, dev, torch, xe, ze, _f32_exact(xs.ravel()), _f32_exact(zs.ravel()), 64, "beyond the focus")
    assert pinned.mean() > 0.5

"""Fuzz of the planar-layer Fermat kernel against the long-double oracle: random media (0..8 interfaces, speeds
1000..6500 m/s), random apertures (sorted / shuffled / clustered / duplicated positions, elements at several depths,
some inside deeper layers), random targets (some above their element -> NaN, offsets up to 100x the depth), sizes that
exercise every workgroup shape (n_e 1..300, n_f 1..5000; every fourth trial a table of 24..130 x 40,000..92,000 — >= 8 rows per
workgroup: the predictor's four-history runs, on regular pitches too — compared on ten of its rows).  Checks NaN masks and
|dt| <= 1e-16 s + 2e-11 t.

    gpurun -- python scripts/fuzz_layers.py [n_trials] [seed] [--taup]
--taup: the tau-p accuracy tier through the device-side sorted entry (rtus_tt_layers_sorted_dev), |dt| <= 1e-16 s + 6e-11 t (the tier's bar).
"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtus
from oracle import cport

TAUP = "--taup" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
trials = int(args[0]) if len(args) > 0 else 200
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 4242)
REL = 6e-11 if TAUP else 2e-11
if TAUP:
    import torch
    from importlib import import_module
    dev_api = import_module("ray-tracing-ultrasound_amd.device")
    t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
worst_rel, worst_abs, solves, t0 = 0.0, 0.0, 0, time.time()
for trial in range(trials):
    n_if = int(rng.integers(0, 9))
    z_if = np.cumsum(rng.uniform(0.001, 0.012, n_if))
    c = rng.uniform(1000.0, 6500.0, n_if + 1)
    if rng.random() < 0.2 and n_if >= 1:
        c[rng.integers(0, n_if + 1)] = c[0]                      # equal speeds in two layers
    n_e = int(rng.choice([1, 2, 3, rng.integers(4, 40), rng.integers(40, 300)]))
    n_f = int(rng.choice([1, rng.integers(2, 70), rng.integers(70, 700), rng.integers(700, 5000)]))
    big = trial % 4 == 3                                          # a table large enough for >= 8 rows per workgroup: the predictor, the
    if big:                                                       # four-history runs and (tau-p tier) the held groups are at work
        n_e = int(rng.integers(24, 130))
        n_f = max(int(rng.integers(40000, 90000)), 2_200_000 // n_e + 64)       # (>= 32,768 wave-solves: 8 rows per workgroup)
    xe = rng.uniform(-0.05, 0.05, n_e)
    k = rng.integers(0, 6 if big else 4)
    if k == 0: xe = np.sort(xe)
    if k == 1: xe = np.resize(np.repeat(xe[: max(1, n_e // 3)], 3), n_e)                # runs of duplicated positions
    if k == 2: xe = np.sort(np.concatenate([rng.normal(-0.01, 1e-5, n_e // 2), rng.normal(0.01, 3e-3, n_e - n_e // 2)]))
    if k == 4: xe = (np.arange(n_e) - n_e / 2) * rng.choice([0.2e-3, 0.6e-3, 2e-3, 5e-3])    # a regular pitch, fine to coarse
    if k == 5: xe = np.concatenate([(np.arange(n_e // 2) - n_e) * 0.3e-3, np.sort(rng.uniform(0.0, 0.04, n_e - n_e // 2))])   # regular, then random
    depth = float(z_if[-1]) if n_if else 0.03
    ze = np.zeros(n_e) if rng.random() < 0.5 else rng.choice([0.0, -0.002, 0.3 * depth, 0.6 * depth], n_e)
    if rng.random() < 0.3:
        ze = np.repeat(rng.uniform(-0.003, 0.5 * depth, (n_e + 7) // 8), 8)[:n_e]          # depth changes every 8 elements
    xf = rng.uniform(-0.06, 0.06, n_f) * (100.0 if rng.random() < 0.1 else 1.0)
    zf = rng.uniform(-0.002, depth + 0.03, n_f)
    if TAUP:
        tt = dev_api.tt_layers_sorted_dev(z_if, c, t64(xe), t64(ze), t64(xf), t64(zf), taup=True).cpu().numpy()
        assert not big or dev_api.rows_per_block(n_e, n_f) >= 8, (n_e, n_f)
    else:
        tt = rtus.travel_time_layers(z_if, c, xe, ze, xf, zf)
    if big:                                                       # the oracle on a sample of the rows (every target of them)
        rows = np.sort(rng.choice(n_e, size=10, replace=False))
        tt, xe, ze = tt[rows], xe[rows], ze[rows]
    ref = cport.tt_layers(z_if, c, xe, ze, xf, zf)
    if not np.array_equal(np.isnan(tt), np.isnan(ref)):
        bad = np.argwhere(np.isnan(tt) != np.isnan(ref))
        print(f"NaN MASK MISMATCH trial {trial} n_if={n_if} n_e={n_e} n_f={n_f} first {bad[:4].tolist()}")
        sys.exit(1)
    m = ~np.isnan(ref)
    if m.any():
        err = np.abs(tt - ref)[m]
        worst_abs = max(worst_abs, float(err.max()))
        worst_rel = max(worst_rel, float((err / ref[m]).max()))
        if np.any(err > 1e-16 + REL * ref[m]):
            i = np.argmax(err / (1e-16 + REL * ref[m]))
            e, f = np.argwhere(m)[i]
            print(f"VALUE MISMATCH trial {trial} n_if={n_if} c={c.tolist()} z_if={z_if.tolist()} elem ({xe[e]}, {ze[e]}) target ({xf[f]}, {zf[f]}): "
                  f"gpu {tt[e, f]!r} oracle {ref[e, f]!r}")
            sys.exit(1)
    solves += int(m.sum())
    if trial % 50 == 49:
        print(f"trial {trial + 1}/{trials}: {solves} solves, worst rel {worst_rel:.2e} abs {worst_abs:.2e} s, {time.time() - t0:.0f} s", flush=True)
print(f"OK{' (tau-p tier, sorted entry)' if TAUP else ''}: {trials} trials, {solves} solves, worst rel {worst_rel:.2e}, worst abs {worst_abs:.2e} s")

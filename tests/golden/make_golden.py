#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Runs only in the build container (where /root/reference is mounted read-only).
The reference's Python is imported, never copied: this script injects the
module globals the reference's ``__main__`` block sets (main_rt.py:449-467,
main_compare.py:450-466) and calls ``shoot_rays(..., plot=False)``
(main_rt.py:337).  Outputs are DATA (inputs + expected outputs) written as
compressed .npz files, plus verbatim copies of the two CSV data files the
reference itself holds as its known answers (compare.csv, database_2.csv).

Usage:  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import shutil
import sys
import warnings

import numpy as np

REF = os.environ.get("RTUS_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import main_rt as M  # noqa: E402  (the reference, imported read-only)

KEYS = ["lens_1_x", "lens_1_z", "pipe_x", "pipe_z",
        "lens_2_x", "lens_2_z", "target_x", "target_z"]


def set_globals(n_rays, r_outer, pipe_offset):
    """main_rt.py:449-467 — constants the helpers read as module globals."""
    M.c1 = np.float64(6400)
    M.c2 = np.float64(1483)
    M.c3 = np.float64(5600)
    M.l0 = np.float64(0.12156646438729327)
    M.h0 = np.float64(0.08843353561270673)
    M.d = M.l0 + M.h0
    M.alpha_max = np.float64(50.62033040986099 * (np.pi / 180))
    M.num_elements = np.int64(64)
    M.pitch = np.float64(0.0006)
    M.num_alpha_points = np.int64(n_rays)
    M.r_outer = np.float64(r_outer)
    M.pipe_offset = np.float64(pipe_offset)


def element_x():
    """main_rt.py:469-474 — 64 elements + virtual centre element at idx 32."""
    x_a = np.arange(M.num_elements, dtype=np.float64) * M.pitch
    x_a = x_a - np.mean(x_a)
    x_aux = list(x_a)
    x_aux.insert(32, np.float64(0.0))
    return np.asarray(x_aux, dtype=np.float64)


def shoot(n_rays, r_outer, pipe_offset, x_tx):
    set_globals(n_rays, r_outer, pipe_offset)
    alpha = np.linspace(-M.alpha_max, M.alpha_max, M.num_alpha_points)
    zf = np.ones((int(n_rays),), dtype=np.float64) * M.d
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = M.shoot_rays(np.float64(x_tx), M.d, zf, alpha, plot=False)
    return alpha, zf, np.stack([np.asarray(res[k], dtype=np.float64) for k in KEYS])


def tof_segments(x_tx, out8):
    """The 4 segment times of main_compare.py:514-517 via the reference's dist()."""
    d = M.d
    t1 = M.dist(x_tx, d, out8[0], out8[1]) / M.c1
    t2 = M.dist(out8[0], out8[1], out8[2], out8[3]) / M.c2
    t3 = M.dist(out8[2], out8[3], out8[4], out8[5]) / M.c2
    t4 = M.dist(out8[4], out8[5], out8[6], out8[7]) / M.c1
    return np.stack([t1, t2, t3, t4])


def main():
    # -- 0. the reference's own known-answer data files --------------------------------
    for name in ("compare.csv", "database_2.csv"):
        shutil.copyfile(os.path.join(REF, name), os.path.join(OUT, name))

    # -- 1. compare config: N=1810, r=0.037, off=0.0038, tx x=0 (main_compare.py:463-481)
    alpha, zf, out8 = shoot(1810, 0.037, 0.0038, 0.0)
    np.savez_compressed(os.path.join(OUT, "compare_cfg.npz"), alpha=alpha, zf=zf,
                        out8=out8, tof4=tof_segments(np.float64(0.0), out8),
                        x_elem=element_x(), r_outer=0.037, pipe_offset=0.0038, x_tx=0.0)
    print("compare_cfg done")

    # -- 2. database_2 sweep: N=905, 10 radii x 21 offsets, tx x=0 (main_rt.py:464-482) --
    geoms, tx_all, full8 = [], [], {}
    keep_full = {(1, -10), (1, 0), (3, 7), (5, -4), (5, 3), (10, 10), (10, -10), (7, 1)}
    for r in range(1, 11):
        for p_o in range(-10, 11):
            alpha, zf, out8 = shoot(905, r * 1e-2, p_o * 1e-3, 0.0)
            geoms.append((float(M.r_outer), float(M.pipe_offset)))
            tof4 = tof_segments(np.float64(0.0), out8)
            # per-ray total in the summation order of main_rt.py:497-500
            tot = ((tof4[0] + tof4[1]) + tof4[2]) + tof4[3]
            tx_all.append(np.stack([out8[6], tot]))
            if (r, p_o) in keep_full:
                full8[f"full_r{r}_o{p_o}"] = out8
        print("sweep r", r)
    np.savez_compressed(os.path.join(OUT, "sweep_cfg.npz"), alpha=alpha,
                        geoms=np.asarray(geoms), target_x_tof=np.asarray(tx_all),
                        x_elem=element_x(), **full8)

    # -- 3. off-centre transmit: all 65 tx x 2 geometries (reference never runs this) -----
    xe = element_x()
    for tag, (r_o, off) in {"a": (0.05, 0.003), "b": (0.037, 0.0038)}.items():
        outs = []
        for x_tx in xe:
            alpha, zf, out8 = shoot(905, r_o, off, x_tx)
            outs.append(out8)
        np.savez_compressed(os.path.join(OUT, f"alltx_{tag}.npz"), alpha=alpha, x_elem=xe,
                            out8=np.asarray(outs), r_outer=r_o, pipe_offset=off)
        print("alltx", tag)

    # -- 4. edge cases ----------------------------------------------------------------------
    edge = {}
    for tag, (r_o, off, x_tx) in {
        "q1nan": (0.01, -0.01, 0.0),      # |x_q| > r  -> NaN tangent (Q1)
        "tir": (0.1, 0.01, 0.0),          # total internal reflection on the way back
        "off0": (0.05, 0.0, 0.0),         # everything retraces to x = 0
        "offtx": (0.03, -0.002, -0.0189),  # off-centre tx, small pipe
    }.items():
        alpha, zf, out8 = shoot(905, r_o, off, x_tx)
        edge[tag] = out8
        edge[tag + "_cfg"] = np.asarray([r_o, off, x_tx])
    # a geometry where rays miss the circle: the reference raises (Q6); record that fact.
    try:
        shoot(905, 0.005, 0.05, 0.0)
        edge["miss_raises"] = np.asarray(0)
    except Exception as e:  # numpy.linalg.LinAlgError: SVD did not converge
        edge["miss_raises"] = np.asarray(1)
        edge["miss_exc"] = np.asarray(type(e).__name__)
    np.savez_compressed(os.path.join(OUT, "edge_cfg.npz"), alpha=alpha, **edge)
    print("edge done")

    # -- 4b. find_line_curve_intersection (main_rt.py:6-168) on hand-made polylines: its corner-case
    #        semantics (first index wins, sign(0) counts, isclose fallbacks, bounds check) as DATA.
    cases = {
        "two_crossings": (0.0, 0.5, [0, 1, 2, 3, 4], [1, 0, 0, 1, 1]),
        "point_on_line": (0.0, 0.0, [0, 1, 2, 3, 4], [1, 0, -1, -1, -1]),
        "no_crossing": (0.0, -1.0, [0, 1, 2, 3, 4], [1, 1, 1, 1, 1]),
        "isclose_point": (0.0, 1.0 - 5e-9, [0, 1, 2, 3, 4], [2, 1, 2, 2, 2]),
        "collinear": (1.0, 0.0, [0, 1, 2], [1e-12, 1.0, 3.0]),
        "vertical_segment": (1.0, 0.0, [1, 1, 2], [2, 0, 0]),
        "steep_line": (250.0, -100.0, [0, 0.2, 0.4, 0.6, 0.8], [0, 1, 3, 2, 0]),
        "first_of_many": (0.3, 0.1, [0, 1, 2, 3, 4, 5], [1, -1, 1, -1, 1, -1]),
        "out_of_bounds_y": (1e9, -1e9, [0.5, 1.5, 2.5], [0.0, 1.0, 2.0]),
    }
    lc = {}
    for tag, (m, b, xc, yc) in cases.items():
        xc, yc = np.asarray(xc, dtype=np.float64), np.asarray(yc, dtype=np.float64)
        xl = np.linspace(float(xc.min()) - 0.15, float(xc.max()) + 0.15, 64)   # the line as the reference samples it (:385-388)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xi, yi = M.find_line_curve_intersection(xl, m * xl + b, xc, yc)
        lc[tag + "_in"] = np.concatenate([[m, b], xc, yc])
        lc[tag + "_out"] = np.asarray([np.nan if xi is None else xi, np.nan if yi is None else yi], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "line_curve_cases.npz"), **lc)
    print("line_curve cases done")

    # -- 5. N-nesting (polyline resolution == ray count, Q3) ------------------------------
    nest = {}
    for n in (181, 1809, 3617):
        alpha, zf, out8 = shoot(n, 0.037, 0.0038, 0.0)
        nest[f"n{n}"] = out8
    np.savez_compressed(os.path.join(OUT, "nest_cfg.npz"), **nest)
    print("nest done")


if __name__ == "__main__":
    main()

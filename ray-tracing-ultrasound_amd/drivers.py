"""The reference's two drivers on top of the accelerated path.

* ``sweep_database`` — the parameter sweep of main_rt.py:464-504 (10 radii x 21 offsets x 65 elements
  -> database_2.csv rows) as ONE batched forward-trace launch + ONE matcher call.
* ``compare_rows`` — the per-ray table of main_compare.py:502-524 (alpha, offset, radius, hitted,
  tof_1..tof_4 -> compare.csv) for one configuration.

CSV text is produced the way the reference's ``csv.writer`` produces it (``str()`` of NumPy scalars:
shortest round-trip float repr, ``True``/``False``, integer ``0`` for "no hit", ``nan``).  Pandas / MSE
comparison against database.csv (main_compare.py:485-500, 526-553) stays the caller's business.

``backend`` is the compute provider; the default is the HIP library (``api``).  Tests inject an
oracle-backed stand-in on machines without a GPU — the product default never does.
"""
import csv
import io

import numpy as np

from . import api


def sweep_geometries(radii_cm=range(1, 11), offsets_mm=range(-10, 11)):
    """(r_outer, pipe_offset) in the reference's loop order and float construction (main_rt.py:464-467)."""
    return np.asarray([[np.float64(r * 1e-2), np.float64(p * 1e-3)] for r in radii_cm for p in offsets_mm],
                      dtype=np.float64)


def sweep_database(*, params=None, n_rays=181 * 5, num_elements=64, pitch=0.0006, element_idx=32,
                   radii_cm=range(1, 11), offsets_mm=range(-10, 11), atol=1e-6, backend=None):
    """main_rt.py:459-504 -> list of rows [elem_idx, offset, radius, hitted, tof_total]."""
    be = backend or api
    p = params or api.Params()
    x_a = api.reference_elements(num_elements, pitch)                      # :469-473
    z_a = np.ones_like(x_a) * p.d                                          # :474
    zf = np.ones((n_rays,), dtype=np.float64) * p.d                        # :477
    alpha = np.linspace(-api.ALPHA_MAX, api.ALPHA_MAX, n_rays)             # :479
    geoms = sweep_geometries(radii_cm, offsets_mm)
    if hasattr(be, "sweep_batch"):                                         # :482-501 for all geometries in one kernel
        b = be.sweep_batch([x_a[element_idx]], [z_a[element_idx]], zf, alpha, x_a, geoms, atol=atol, params=p)
        hit, tof = b["hit"][:, 0], b["tof_hit"][:, 0]
    else:                                                                  # a backend with the two reference-shaped calls only
        b = be.shoot_batch([x_a[element_idx]], [z_a[element_idx]], zf, alpha, geoms, params=p,
                           want=("tof", "land_x"))                         # :482, all geometries at once
        hit, tof, _ = be.match_elements(b["land_x"][:, 0], b["tof"][:, 0], x_a, atol=atol)   # :487-501
    rows = []
    for g in range(geoms.shape[0]):
        for e in range(x_a.size):
            rows.append([e, np.float64(geoms[g, 1]), np.float64(geoms[g, 0]), bool(hit[g, e]),
                         np.float64(tof[g, e]) if hit[g, e] else 0])      # :489-493 (0 is an int)
    return rows


def compare_rows(*, params=None, n_rays=181 * 10, num_elements=64, pitch=0.0006, element_idx=32, atol=1e-4,
                 backend=None):
    """main_compare.py:463-481, 508-524 -> (header, rows [alpha, offset, radius, hitted, tof_1..4])."""
    be = backend or api
    p = params or api.Params(r_outer=0.037, pipe_offset=0.0038)            # main_compare.py:465-466
    x_a = api.reference_elements(num_elements, pitch)
    zf = np.ones((n_rays,), dtype=np.float64) * p.d
    alpha = np.linspace(-api.ALPHA_MAX, api.ALPHA_MAX, n_rays)
    b = be.shoot_batch([x_a[element_idx]], [p.d], zf, alpha, params=p, want=("tof4", "land_x"))
    tof4 = b["tof4"][0, 0]
    hitted = be.ray_hits(b["land_x"][0, 0], x_a, atol=atol)                # :518-521
    header = ["alpha", "offset", "radius", "hitted", "tof_1", "tof_2", "tof_3", "tof_4"]   # :483
    rows = [[alpha[r], np.float64(p.pipe_offset), np.float64(p.r_outer), bool(hitted[r]),
             tof4[0, r], tof4[1, r], tof4[2, r], tof4[3, r]] for r in range(n_rays)]
    return header, rows


def rows_to_csv(rows, header=None) -> str:
    """Text exactly as the reference's csv.writer(newline='') emits it (\\r\\n line ends)."""
    buf = io.StringIO(newline="")
    w = csv.writer(buf)
    if header:
        w.writerow(header)
    w.writerows(rows)
    return buf.getvalue()


def compare_against_database(database_csv, *, params=None, backend=None, **kw):
    """The tail of main_compare.py (:485-500, 526-553) without pandas: pick the (offset, radius) group of
    ``database_csv`` nearest to this configuration, sum the four segment times per ray (NaN if any is NaN),
    subtract this run's sums ray by ray and return (individual_errors, num_hitted, mse) with
    mse = sum(err^2 over non-NaN) / num_hitted, exactly as the reference prints them.

    ``database.csv`` is NOT in the reference checkout (.MISSING_LARGE_BLOBS): a missing file raises
    FileNotFoundError with that explanation instead of the reference's bare pandas error.
    """
    import os
    if database_csv is None or not os.path.exists(database_csv):
        raise FileNotFoundError(f"{database_csv!r} not found — the reference's database.csv is absent from its checkout "
                                "(.MISSING_LARGE_BLOBS); pass a CSV with columns alpha,offset,radius,hitted,tof_1..tof_4")
    p = params or api.Params(r_outer=0.037, pipe_offset=0.0038)
    with open(database_csv, newline="") as fh:
        rd = csv.DictReader(fh)
        db = [(float(r["offset"]), float(r["radius"]), [float(r[f"tof_{k}"]) for k in (1, 2, 3, 4)]) for r in rd]
    off = np.array([d[0] for d in db]); rad = np.array([d[1] for d in db])
    off_c = off[np.argmin(np.abs(off - p.pipe_offset))]                    # :489-493
    rad_c = rad[np.argmin(np.abs(rad - p.r_outer))]
    sel = (off == off_c) & (rad == rad_c)                                  # :495-500
    t_db = np.array([d[2] for d, s_ in zip(db, sel) if s_], dtype=np.float64)
    header, rows = compare_rows(params=p, backend=backend, **kw)
    t_cmp = np.array([r[4:8] for r in rows], dtype=np.float64)
    if t_db.shape != t_cmp.shape:
        raise ValueError(f"database group has {t_db.shape[0]} rays, this run {t_cmp.shape[0]} (main_compare.py:544 "
                         "subtracts them element-wise)")
    sums_db, sums_cmp = t_db.sum(axis=1), t_cmp.sum(axis=1)                # NaN if any segment is NaN (:526-534)
    err = sums_db - sums_cmp                                               # :544
    num_hitted = int(sum(bool(r[3]) for r in rows))                        # :540
    mse = float(np.sum(np.square(err[~np.isnan(err)])) / num_hitted) if num_hitted else float("nan")   # :550-551
    return err, num_hitted, mse

"""NumPy restatement of the reference's forward trace WITH ITS LOOP STRUCTURE — TEST INFRASTRUCTURE (see oracle/__init__.py).

rt_numpy.py lays the O(N^2) polyline scan out as 2-D array operations (~80x kinder to the CPU than the reference is); this
module keeps what makes the reference slow, because that is the CPU path a user of main_rt.py actually runs:
a vectorised prologue and epilogue around a Python `for ray in range(N)` loop (main_rt.py:384-393) that, per ray, samples the
reflected line on an N-point abscissa grid, hands it to a line-vs-polyline routine which re-derives slope and intercept from
the samples, subtracts the line from all N polyline points and looks for the first sign change (main_rt.py:6-168: ~10 NumPy
calls on N-element arrays per ray).  The one deliberate difference: the reference re-fits the line with np.polyfit (LAPACK
lstsq, main_rt.py:73); here the two end samples give slope and intercept (<= 1.3e-18 s effect, and polyfit alone is
half of the reference's run time — this leg is therefore still ~2x kinder than the reference).  bench.py times it beside the
GPU (`cpu_ref_path.numpy_loop_port_*`); tests/test_oracle_golden.py pins it to the reference's outputs."""
import numpy as np

from .rt_numpy import PI_2, _refract, lens_point_tangent


def line_meets_polyline(x_line, z_line, x_curve, z_curve):
    """main_rt.py:6-168 for a non-vertical sampled line -> (x, z) of the first crossing in index order, or (None, None)."""
    if x_line.size < 2 or x_curve.size < 2:
        raise ValueError("Curve needs at least two points.")                       # :24-29
    m = (z_line[-1] - z_line[0]) / (x_line[-1] - x_line[0])                        # :73-74 (polyfit in the reference)
    b = z_line[0] - m * x_line[0]
    gap = z_curve - (m * x_curve + b)                                              # :78-79
    flips = np.where(np.diff(np.sign(gap)) != 0)[0]                                # :82
    if flips.size == 0:                                                            # :84-96
        on = np.where(np.isclose(gap, 0.0))[0]
        return (x_curve[on[0]], z_curve[on[0]]) if on.size else (None, None)
    j = flips[0]                                                                   # :99
    x1, z1, x2, z2 = x_curve[j], z_curve[j], x_curve[j + 1], z_curve[j + 1]
    if np.isclose(x1, x2):                                                         # :110-123 vertical chord
        z = m * x1 + b
        return (x1, z) if min(z1, z2) - 1e-9 <= z <= max(z1, z2) + 1e-9 else (None, None)
    m_c = (z2 - z1) / (x2 - x1)                                                    # :127-128
    b_c = z1 - m_c * x1
    if np.isclose(m, m_c):                                                         # :131-144 parallel / collinear
        if np.isclose(b, b_c):
            x = (x1 + x2) / 2.0
            return x, m * x + b
        return None, None
    x = (b_c - b) / (m - m_c)                                                      # :147-150
    z = m * x + b
    inside = (min(x1, x2) - 1e-9 <= x <= max(x1, x2) + 1e-9) and (min(z1, z2) - 1e-9 <= z <= max(z1, z2) + 1e-9)   # :153-168
    return (x, z) if inside else (None, None)


def shoot(x_a, z_a, z_f, alpha, r_outer, pipe_offset, c1=6400.0, c2=1483.0, l0=0.12156646438729327,
          h0=0.08843353561270673, d=None):
    """main_rt.py:337-405 -> out8 [8, n] in the reference's key order; the per-ray loop is the reference's."""
    d = l0 + h0 if d is None else d
    alpha = np.asarray(alpha, dtype=np.float64)
    z_f = np.asarray(z_f, dtype=np.float64)
    n = alpha.size
    with np.errstate(invalid="ignore", divide="ignore"):
        x_p, z_p, dz, dx = lens_point_tangent(alpha, c1, c2, l0, h0, d)                       # :338, 344
        phi_pq = _refract(np.arctan2(z_a - z_p, x_a - x_p), np.arctan2(dz, dx), c2 / c1)      # :341-345
        a = np.tan(phi_pq)                                                                    # :348-349
        bq = z_p - a * x_p
        A, B = a * a + 1, 2 * (a * bq - pipe_offset)                                          # :351-357
        C = pipe_offset * pipe_offset + bq * bq - r_outer * r_outer
        sq = np.sqrt(B * B - 4 * A * C)
        x1, x2 = (-B + sq) / (2 * A), (-B - sq) / (2 * A)
        z1, z2 = a * x1 + bq, a * x2 + bq
        up = z1 > z2                                                                          # :362-364
        x_q, z_q = np.where(up, x1, x2), np.where(up, z1, z2)
        phi_s = np.arctan(-x_q / np.sqrt(r_outer * r_outer - x_q * x_q))                      # :237-238, 287
        phi_l = phi_s - PI_2 - (phi_pq - (phi_s + PI_2))                                      # :289-291
        a_l = np.tan(phi_l)                                                                   # :375-376
        b_l = z_q - a_l * x_q
        x_i, z_i = np.full(n, np.nan), np.full(n, np.nan)
        for ray in range(n):                                                                  # :384-393 THE hot loop
            if not (np.isfinite(a_l[ray]) and np.isfinite(b_l[ray])):
                continue                                                                      # (the reference ends in None or raises, Q6)
            xx = np.linspace(x_q[ray] - 0.15, x_q[ray] + 0.15, n)                             # :385 (n samples of the line)
            zz = a_l[ray] * xx + b_l[ray]
            xi, zi = line_meets_polyline(xx, zz, x_p, z_p)
            if xi is not None:
                x_i[ray], z_i[ray] = xi, zi
        _, _, dzi, dxi = lens_point_tangent(np.arctan2(x_i, z_i), c1, c2, l0, h0, d)          # :396-397
        a3 = np.tan(_refract(phi_l, np.arctan2(dzi, dxi), c1 / c2))                           # :398-401
        x_in = (z_f - (z_i - a3 * x_i)) / a3                                                  # :402-404
    return np.stack([x_p, z_p, x_q, z_q, x_i, z_i, x_in, z_f.copy()])

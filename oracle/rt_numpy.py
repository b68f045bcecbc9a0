"""NumPy restatement of the reference's forward trace — TEST INFRASTRUCTURE (see oracle/__init__.py).

Same algorithm and the same O(N^2) polyline scan as main_rt.py:337-405 (every ray tests every
polyline point for the first sign change), but laid out as 2-D array operations over a chunk of
rays instead of the reference's per-ray Python loop + np.polyfit.  Used (a) as a second,
independently written checker next to rt_oracle.c and (b) as the "NumPy CPU path" leg bench.py
reports beside the GPU.  Pinned by tests/test_oracle_golden.py against the reference's outputs.
"""
import numpy as np

PI_2 = np.pi / 2


def lens_point_tangent(alpha, c1, c2, l0, h0, d):
    """main_rt.py:180-234: lens polar radius h(alpha), point (x, z) and tangent (dz, dx)."""
    T = l0 / c1 + h0 / c2
    A = (c1 * c1) / (c2 * c2) - 1
    C = (c1 * c1) * (T * T) - d * d
    ca, sa = np.cos(alpha), np.sin(alpha)
    B = 2 * d * ca - 2 * T * (c1 * c1) / c2                        # :184
    h = (-B - np.sqrt(B * B - 4 * A * C)) / (2 * A)                # :171-177, root [1]
    Bd = (-2 * T * (c1 * c1)) / c2 + (2 * d) * ca                  # :203
    dB = -(2 * d) * sa
    dh = (-1 / (2 * A)) * (dB + (1 / (2 * np.sqrt(Bd * Bd - 4 * A * C))) * (2 * Bd * dB))   # :199-212
    return h * sa, h * ca, dh * ca - h * sa, dh * sa + h * ca


def _refract(phi_in, phi_slope, ratio):
    """main_rt.py:276-278 with the slope angle given."""
    with np.errstate(invalid="ignore"):
        return phi_slope - PI_2 + np.arcsin(ratio * np.sin(phi_in - (phi_slope + PI_2)))


def first_crossings(m, b, xc, zc, chunk=256):
    """main_rt.py:78-168 for many lines at once -> (xi, zi) with NaN for the reference's None."""
    n_l, n = m.size, xc.size
    xi = np.full(n_l, np.nan)
    zi = np.full(n_l, np.nan)
    for s in range(0, n_l, chunk):
        mm, bb = m[s:s + chunk, None], b[s:s + chunk, None]
        with np.errstate(invalid="ignore"):
            D = zc[None, :] - (mm * xc[None, :] + bb)              # :78-79
            chg = np.diff(np.sign(D), axis=1) != 0                 # :82 (NaN != 0 is True)
        has = chg.any(axis=1)
        idx = chg.argmax(axis=1)                                   # :99 first change
        on = np.abs(D) <= 1e-8                                     # :86 isclose(diffs, 0)
        for i in np.nonzero(~has & on.any(axis=1))[0]:             # :84-90
            j = on[i].argmax()
            xi[s + i], zi[s + i] = xc[j], zc[j]
        rows = np.nonzero(has)[0]
        j = idx[rows]
        x1, y1, x2, y2 = xc[j], zc[j], xc[j + 1], zc[j + 1]
        ml, bl = m[s + rows], b[s + rows]
        with np.errstate(invalid="ignore", divide="ignore"):
            vert = np.isclose(x1, x2)                              # :110
            m_seg = (y2 - y1) / (x2 - x1)                          # :127-128
            b_seg = y1 - m_seg * x1
            par = np.isclose(ml, m_seg)                            # :131
            col = par & np.isclose(bl, b_seg)                      # :133
            x = np.where(vert, x1, np.where(col, (x1 + x2) / 2.0, (b_seg - bl) / (ml - m_seg)))   # :112,137,147
            y = ml * x + bl                                        # :113,138,150
            inx = (x >= np.minimum(x1, x2) - 1e-9) & (x <= np.maximum(x1, x2) + 1e-9)
            iny = (y >= np.minimum(y1, y2) - 1e-9) & (y <= np.maximum(y1, y2) + 1e-9)
            ok = np.where(vert, iny, np.where(par, col, inx & iny))  # :116, 133-144, 157-158
        xi[s + rows] = np.where(ok, x, np.nan)
        zi[s + rows] = np.where(ok, y, np.nan)
    return xi, zi


def shoot(x_a, z_a, z_f, alpha, r_outer, pipe_offset, c1=6400.0, c2=1483.0, l0=0.12156646438729327,
          h0=0.08843353561270673, d=None):
    """main_rt.py:337-405 -> out8 [8, n] in the reference's key order."""
    d = l0 + h0 if d is None else d
    alpha = np.asarray(alpha, dtype=np.float64)
    z_f = np.asarray(z_f, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        x_p, z_p, dz, dx = lens_point_tangent(alpha, c1, c2, l0, h0, d)              # :338, 344
        phi_pq = _refract(np.arctan2(z_a - z_p, x_a - x_p), np.arctan2(dz, dx), c2 / c1)   # :341-345
        a = np.tan(phi_pq)                                                           # :348-349
        bq = z_p - a * x_p
        A, B = a * a + 1, 2 * (a * bq - pipe_offset)                                 # :351-357
        C = pipe_offset * pipe_offset + bq * bq - r_outer * r_outer
        sq = np.sqrt(B * B - 4 * A * C)
        x1, x2 = (-B + sq) / (2 * A), (-B - sq) / (2 * A)                            # :171-177
        z1, z2 = a * x1 + bq, a * x2 + bq
        up = z1 > z2                                                                 # :362-364
        x_q, z_q = np.where(up, x1, x2), np.where(up, z1, z2)
        phi_s = np.arctan(-x_q / np.sqrt(r_outer * r_outer - x_q * x_q))             # :237-238, 287
        phi_l = phi_s - PI_2 - (phi_pq - (phi_s + PI_2))                             # :289-291
        m = np.tan(phi_l)                                                            # :375-376
        bl = z_q - m * x_q
        xi, zi = first_crossings(m, bl, x_p, z_p)                                    # :384-393
        _, _, dzi, dxi = lens_point_tangent(np.arctan2(xi, zi), c1, c2, l0, h0, d)   # :396-397
        a3 = np.tan(_refract(phi_l, np.arctan2(dzi, dxi), c1 / c2))                  # :398-401
        x_in = (z_f - (zi - a3 * xi)) / a3                                           # :402-404
    return np.stack([x_p, z_p, x_q, z_q, xi, zi, x_in, z_f.copy()])

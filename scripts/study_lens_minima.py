"""CPU study behind the lens kernel's "suspect lane" rule ("Round 4: the least time everywhere"; NumPy only, no GPU):
for (element, target) pairs below the reference lens, T(alpha) on a dense grid of alpha ->
  (1) where T has more than one local minimum over [-alpha_max, alpha_max], how large is g' = d2T/dalpha2 at its INTERIOR minima?
  (2) how small is g' at the minimum of pairs with ONE minimum (tables away from the focus)?
  (3) how fast does g' at the followed minimum move with the element position?
The kernel's constants gp_min (7.5e-6 s/rad^2 = twice the bound of (1)) and gp_dx (3.9e-4 s/rad^2 per metre, (3)) come from here.
    python scripts/study_lens_minima.py          (~2 minutes, ~3 GB)
"""
import numpy as np

c1, c2, l0, h0 = 6400.0, 1483.0, 0.12156646438729327, 0.08843353561270673     # main_rt.py:449-457
d = l0 + h0
AM = np.deg2rad(50.62033040986099)


def lens_point(al):                                                              # main_rt.py:171-189, 217-223
    T = l0 / c1 + h0 / c2
    A = c1 * c1 / (c2 * c2) - 1
    B = 2 * d * np.cos(al) - 2 * T * c1 * c1 / c2
    C = c1 * c1 * T * T - d * d
    h = (-B - np.sqrt(B * B - 4 * A * C)) / (2 * A)
    return h * np.sin(al), h * np.cos(al)


al = np.linspace(-AM, AM, 4001)
px, pz = lens_point(al)
da = al[1] - al[0]


def landscape(xe, xf, zf):
    """T[element, target, alpha]"""
    return (np.hypot(px[None, None, :] - xe[:, None, None], pz[None, None, :] - d) / c1 +
            np.hypot(px[None, None, :] - xf[None, :, None], pz[None, None, :] - zf[None, :, None]) / c2)


def minima(name, xe, xf, zf, chunk=64):
    worst_multi, least_single, n_multi, n_all, gap = 0.0, np.inf, 0, 0, 0.0
    for i in range(0, xf.size, chunk):
        T = landscape(xe, xf[i:i + chunk], zf[i:i + chunk])
        dT = np.diff(T, axis=-1)
        lm = (dT[..., :-1] < 0) & (dT[..., 1:] >= 0)                             # interior local minimum at index k + 1
        nmin = lm.sum(-1) + (dT[..., 0] >= 0) + (dT[..., -1] < 0)               # + minima pinned at the ends
        g2 = (T[..., 2:] - 2 * T[..., 1:-1] + T[..., :-2]) / da ** 2
        multi = nmin > 1
        n_all += multi.size
        n_multi += int(multi.sum())
        if multi.any():
            worst_multi = max(worst_multi, float(np.where(lm & multi[..., None], g2, 0).max()))
            Tm = np.concatenate([np.where(lm, T[..., 1:-1], np.inf), np.where(dT[..., :1] >= 0, T[..., :1], np.inf),
                                 np.where(dT[..., -1:] < 0, T[..., -1:], np.inf)], -1)
            s = np.sort(Tm, axis=-1)
            gap = max(gap, float((s[..., 1] - s[..., 0])[multi].max()))
        least_single = min(least_single, float(np.where(lm & ~multi[..., None], g2, np.inf).min()))
    print(f"{name}: {n_all} pairs, {n_multi / n_all:.1%} with more than one local minimum; largest g' at an interior minimum of those "
          f"{worst_multi:.3e}; least g' at the minimum of the others {least_single:.3e}; largest T gap between competing minima {gap:.2e} s")


def below_lens(xl, zl, margin=0.003):
    rr, aa = np.hypot(xl, zl), np.arctan2(xl, zl)
    hx, hz = lens_point(np.clip(aa, -AM, AM))
    return (rr < np.hypot(hx, hz) - margin) & ((np.abs(aa) < AM) | (zl < 0.02))


if __name__ == "__main__":
    xe = np.linspace(-0.02, 0.02, 81)
    xs, zs = np.meshgrid(np.linspace(-0.004, 0.004, 24), np.linspace(0.03, 0.07, 24))
    minima("BASELINE configs[3] region", (np.arange(0, 1024, 16) - 511.5) * 0.3e-4, xs.ravel(), zs.ravel())
    xl, zl = np.meshgrid(np.linspace(-0.006, 0.006, 121), np.linspace(-0.006, 0.012, 181))
    minima("around the focus (+-6 mm x -6 .. 12 mm)", xe, xl.ravel(), zl.ravel())
    xl, zl = np.meshgrid(np.linspace(-0.03, 0.03, 61), np.linspace(-0.03, 0.085, 116))
    ok = below_lens(xl.ravel(), zl.ravel())
    minima("the water below the lens", xe, xl.ravel()[ok], zl.ravel()[ok])
    # (3) g' at the followed (global) minimum along the element axis, by target depth
    xe = np.linspace(-0.024, 0.024, 97)
    for zc in (0.002, 0.006, 0.01, 0.015, 0.02, 0.03, 0.05, 0.07):
        T = landscape(xe, np.linspace(-0.006, 0.006, 13), np.full(13, zc))
        jm = np.argmin(T, -1)
        interior = (jm > 2) & (jm < al.size - 3)
        g2 = (T[..., 2:] - 2 * T[..., 1:-1] + T[..., :-2]) / da ** 2
        g = np.where(interior, np.take_along_axis(g2, (np.clip(jm, 1, al.size - 2) - 1)[..., None], -1)[..., 0], np.nan)
        print(f"z = {zc * 1e3:4.0f} mm: g' at the minimum {np.nanmin(g):.2e} .. {np.nanmax(g):.2e} s/rad^2; "
              f"max |dg'/dx_e| {np.nanmax(np.abs(np.diff(g, axis=0))) / (xe[1] - xe[0]):.2e} per metre (1/c1 = {1 / c1:.2e})")

"""A script laid out the way the reference's main_rt.py sweep is (main_rt.py:449-504): constants as globals of
__main__, two nested loops that REBIND the globals r_outer / pipe_offset, one shoot_rays call per geometry, then a
per-element search for the first ray that lands on the element — but calling rtus.shoot_rays.  No rtus.Params
anywhere: the library reads the constants from this module's globals at every call, as the reference's helpers do.
Run as __main__ by tests/test_gpu_dropin_script.py; writes the rows of the reference's database_2.csv."""
import csv
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rtus  # noqa: E402

shoot_rays = rtus.shoot_rays


def dist(x1, z1, x2, z2):
    return np.sqrt((x1 - x2) ** 2 + (z1 - z2) ** 2)


if __name__ == "__main__":
    c1 = np.float64(6400)
    c2 = np.float64(1483)
    l0 = np.float64(0.12156646438729327)
    h0 = np.float64(0.08843353561270673)
    d = l0 + h0
    alpha_max = np.float64(50.62033040986099 * (np.pi / 180))
    num_elements = np.int64(64)
    pitch = np.float64(0.0006)
    num_alpha_points = np.int64(181 * 5)

    x_a = np.arange(num_elements, dtype=np.float64) * pitch
    x_a = np.insert(x_a - np.mean(x_a), 32, np.float64(0.0))
    z_a = np.ones_like(x_a) * d
    zf = np.ones((num_alpha_points,), dtype=np.float64) * d
    alpha = np.linspace(-alpha_max, alpha_max, num_alpha_points)
    element_idx = 32

    with open(sys.argv[1], "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["elem_idx", "offset", "radius", "hitted", "tof_total"])
        for r_cm in range(1, 11):
            r_outer = np.float64(r_cm * 1e-2)                 # rebinding the module global, as the reference does
            for off_mm in range(-10, 11):
                pipe_offset = np.float64(off_mm * 1e-3)
                res = shoot_rays(x_a[element_idx], z_a[element_idx], zf, alpha, plot=False)
                for e in range(x_a.size):
                    rays = np.nonzero(np.isclose(res["target_x"], x_a[e], atol=1e-6))[0]
                    if rays.size:
                        k = rays[0]
                        tof = (dist(x_a[element_idx], z_a[element_idx], res["lens_1_x"][k], res["lens_1_z"][k]) / c1
                               + dist(res["lens_1_x"][k], res["lens_1_z"][k], res["pipe_x"][k], res["pipe_z"][k]) / c2
                               + dist(res["pipe_x"][k], res["pipe_z"][k], res["lens_2_x"][k], res["lens_2_z"][k]) / c2
                               + dist(res["lens_2_x"][k], res["lens_2_z"][k], res["target_x"][k], res["target_z"][k]) / c1)
                        w.writerow([e, pipe_offset, r_outer, True, tof])
                    else:
                        w.writerow([e, pipe_offset, r_outer, False, 0])

"""A script laid out the way the reference's main_compare.py is (constants assigned as globals of __main__,
main_compare.py:450-466; one shoot_rays call, :481; per-ray CSV rows, :508-524) — but calling rtus.shoot_rays.
Run as __main__ by tests/test_gpu_dropin_script.py to prove the drop-in claim: NO rtus.Params is passed, the
library picks the constants up from this module's globals exactly like the reference's helpers do."""
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
    num_alpha_points = np.int64(181 * 10)
    r_outer = np.float64(0.037)
    pipe_offset = np.float64(0.0038)

    x_a = np.arange(num_elements, dtype=np.float64) * pitch
    x_a = np.insert(x_a - np.mean(x_a), 32, np.float64(0.0))
    z_a = np.ones_like(x_a) * d
    zf = np.ones((num_alpha_points,), dtype=np.float64) * d
    alpha = np.linspace(-alpha_max, alpha_max, num_alpha_points)

    element_idx = 32
    results = shoot_rays(x_a[element_idx], z_a[element_idx], zf, alpha, plot=False)

    with open(sys.argv[1], "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["alpha", "offset", "radius", "hitted", "tof_1", "tof_2", "tof_3", "tof_4"])
        for ray in range(num_alpha_points):
            row = [alpha[ray], pipe_offset, r_outer, False,
                   dist(x_a[element_idx], z_a[element_idx], results["lens_1_x"][ray], results["lens_1_z"][ray]) / c1,
                   dist(results["lens_1_x"][ray], results["lens_1_z"][ray], results["pipe_x"][ray], results["pipe_z"][ray]) / c2,
                   dist(results["pipe_x"][ray], results["pipe_z"][ray], results["lens_2_x"][ray], results["lens_2_z"][ray]) / c2,
                   dist(results["lens_2_x"][ray], results["lens_2_z"][ray], results["target_x"][ray], results["target_z"][ray]) / c1]
            row[3] = bool(np.any(np.isclose(results["target_x"][ray], x_a, atol=1e-4)))
            w.writerow(row)

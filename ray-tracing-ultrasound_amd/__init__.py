"""rtus — MI355X-native travel-time ray tracer (drop-in for the hot path of
edudscrc/ray-tracing-ultrasound: main_rt.py's shoot_rays + element matcher).

The directory is called ``ray-tracing-ultrasound_amd``; import it as ``import rtus`` (the alias
module at the repo root) or ``importlib.import_module("ray-tracing-ultrasound_amd")``.
"""
from ._lib import EXPORTS, LIB_PATH, Lens, RtusError, build, lib  # noqa: F401
from .api import (ALPHA_MAX, KEYS, Params, configure, match_elements, ray_hits,  # noqa: F401
                  reference_elements, shoot_batch, shoot_rays, travel_time_layers, travel_time_lens,
                  fmc_table_layers, solve_travel_times, focal_delays, tfm_image, sweep_batch)

__all__ = ["shoot_rays", "shoot_batch", "sweep_batch", "match_elements", "ray_hits", "travel_time_layers", "travel_time_lens", "fmc_table_layers", "solve_travel_times", "focal_delays", "tfm_image", "Params",
           "configure", "reference_elements", "ALPHA_MAX", "KEYS", "build", "lib", "RtusError"]

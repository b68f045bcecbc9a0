"""ctypes loader for librtus.so (the C ABI in include/rtus.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded this module raises,
loudly.  ``oracle/`` is never imported from here.
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librtus.so")
if os.environ.get("RTUS_LIB"):           # experiment builds (scripts/build_variant.sh, interleaved A/B runs): never silently
    LIB_PATH = os.environ["RTUS_LIB"]
    print(f"[rtus] RTUS_LIB is set: loading {LIB_PATH} instead of the in-tree librtus.so", file=sys.stderr)
CSRC = os.path.join(HERE, "csrc")

_lib = None


class RtusError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__(f"{what}: rtus status {status} ({_strerror(status)})")


class Lens(C.Structure):
    """rtus_lens — replaces the reference's module globals c1, c2, l0, h0, d (main_rt.py:449-455)."""
    _fields_ = [("c1", C.c_double), ("c2", C.c_double), ("l0", C.c_double), ("h0", C.c_double),
                ("d", C.c_double)]


def build(force: bool = False) -> str:
    """Compile librtus.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def _strerror(status):
    try:
        return lib().rtus_strerror(int(status)).decode()
    except Exception:  # pragma: no cover
        return "?"


def lib():
    """Load librtus.so once; declare every prototype of include/rtus.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ray-tracing-ultrasound_amd/csrc`. There is no CPU fallback.")
    # ONE HIP runtime per process.  The NumPy call surface (api.py, drivers.py) needs no PyTorch and does not import it; but a
    # process may import torch LATER (device.py / dist.py do), and PyTorch's wheel carries its own libamdhip64.so (SONAME
    # libamdhip64.so.7, found by libtorch_hip through RPATH=$ORIGIN under the NAME libamdhip64.so — which glibc does not match
    # against an already loaded /opt/rocm copy).  Two runtimes in one process would hand torch's device pointers and streams to a
    # library bound to the other one.  So when torch is installed its copy of the runtime is loaded first — a dlopen, not an
    # import: librtus.so's NEEDED libamdhip64.so.7 then resolves to it by SONAME, and a later `import torch` finds the same file
    # already mapped.  Without torch the system runtime (/opt/rocm/lib) is used.  RTUS_SYSTEM_HIP=1 skips this.
    if "torch" not in sys.modules and os.environ.get("RTUS_SYSTEM_HIP", "0") != "1":
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(hip):
                C.CDLL(hip, mode=C.RTLD_GLOBAL)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_int
    L.rtus_strerror.argtypes = [ip]
    L.rtus_strerror.restype = C.c_char_p
    L.rtus_version.restype = ip
    L.rtus_last_hip_error.restype = ip
    L.rtus_device_count.argtypes = [C.POINTER(C.c_int)]
    L.rtus_release.argtypes = [ip]
    L.rtus_release.restype = ip
    L.rtus_selftest.argtypes = [C.POINTER(Lens), ip, C.c_longlong, C.POINTER(C.c_ulonglong), ip]
    L.rtus_selftest.restype = ip
    L.rtus_shoot_workspace_bytes.argtypes = [ip]
    L.rtus_shoot_workspace_bytes.restype = C.c_size_t
    LP = C.POINTER(Lens)
    L.rtus_shoot_dev.argtypes = [LP, dp, ip, dp, dp, ip, dp, dp, ip, dp, dp, dp, dp, vp, vp, C.c_size_t, C.c_uint, vp]
    L.rtus_shoot.argtypes = [LP, dp, ip, dp, dp, ip, dp, dp, ip, dp, dp, dp, dp, vp, C.c_uint, ip]
    L.rtus_match_dev.argtypes = [dp, dp, ip, ip, dp, ip, C.c_double, C.c_double, vp, vp, dp, vp]
    L.rtus_match.argtypes = [dp, dp, ip, ip, dp, ip, C.c_double, C.c_double, vp, vp, dp, ip]
    L.rtus_sweep_workspace_bytes.argtypes = [ip, ip, ip, ip]
    L.rtus_sweep_workspace_bytes.restype = C.c_size_t
    L.rtus_sweep_dev.argtypes = [LP, dp, ip, dp, dp, ip, dp, dp, ip, dp, ip, C.c_double, C.c_double, vp, vp, dp, dp, dp, vp, C.c_size_t,
                                 C.c_uint, vp]
    L.rtus_sweep.argtypes = [LP, dp, ip, dp, dp, ip, dp, dp, ip, dp, ip, C.c_double, C.c_double, vp, vp, dp, dp, dp, C.c_uint, ip]
    L.rtus_sweep_dev.restype = L.rtus_sweep.restype = ip
    L.rtus_ray_hits_dev.argtypes = [dp, ip, ip, dp, ip, C.c_double, C.c_double, vp, vp]
    L.rtus_ray_hits.argtypes = [dp, ip, ip, dp, ip, C.c_double, C.c_double, vp, ip]
    L.rtus_tt_layers_dev.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, vp, vp]
    L.rtus_tt_layers.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, vp, ip]
    L.rtus_tt_layers_batch_dev.argtypes = [dp, dp, ip, dp, dp, ip, C.c_longlong, dp, dp, ip, C.c_longlong, dp,
                                           C.c_longlong, ip, vp]
    L.rtus_tt_layers_batch_dev.restype = ip
    L.rtus_tt_lens_dev.argtypes = [LP, C.c_double, C.c_double, dp, dp, ip, dp, dp, ip, dp, dp, vp]
    L.rtus_tt_lens.argtypes = [LP, C.c_double, C.c_double, dp, dp, ip, dp, dp, ip, dp, dp, ip]
    L.rtus_tt_lens_f32_dev.argtypes = L.rtus_tt_lens_dev.argtypes
    L.rtus_tt_lens_f32.argtypes = L.rtus_tt_lens.argtypes
    L.rtus_focal_delays_dev.argtypes = [dp, ip, ip, dp, vp]
    L.rtus_focal_delays.argtypes = [dp, ip, ip, dp, ip]
    L.rtus_tfm_dev.argtypes = [dp, ip, ip, ip, C.c_double, C.c_double, dp, dp, ip, dp, vp]
    L.rtus_tfm.argtypes = [dp, ip, ip, ip, C.c_double, C.c_double, dp, dp, ip, dp, ip]
    for name in ("rtus_focal_delays_dev", "rtus_focal_delays", "rtus_tfm_dev", "rtus_tfm"):
        getattr(L, name).restype = ip
    L.rtus_solve_workspace_bytes.argtypes = [ip, ip, ip, ip]
    L.rtus_solve_workspace_bytes.restype = C.c_size_t
    L.rtus_solve_dev.argtypes = [LP, dp, ip, dp, dp, ip, dp, ip, dp, ip, C.c_double, dp, dp, dp, dp, vp, vp, C.c_size_t,
                                 C.c_uint, vp]
    L.rtus_solve.argtypes = [LP, dp, ip, dp, dp, ip, dp, ip, dp, ip, C.c_double, dp, dp, dp, dp, vp, C.c_uint, ip]
    L.rtus_solve_dev.restype = ip
    L.rtus_solve.restype = ip
    for name in ("rtus_tt_lens_dev", "rtus_tt_lens", "rtus_tt_lens_f32_dev", "rtus_tt_lens_f32"):
        getattr(L, name).restype = ip
    ll = C.c_longlong
    L.rtus_tt_layers_sort_workspace_bytes.argtypes = [ip]
    L.rtus_tt_layers_sort_workspace_bytes.restype = C.c_size_t
    L.rtus_tt_layers_sorted_dev.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, vp, C.c_size_t, C.c_uint, vp]
    L.rtus_tt_layers_sorted_dev.restype = ip
    L.rtus_table_rows_per_block.argtypes = [ll, ip, ip]
    L.rtus_table_rows_per_block.restype = ip
    L.rtus_shard_rows.argtypes = [ll, ip, ip, ip]
    L.rtus_shard_rows.restype = ll
    L.rtus_tt_layers_rows_dev.argtypes = [dp, dp, ip, dp, dp, ip, ll, ll, dp, dp, ip, dp, C.c_uint, vp]
    L.rtus_tt_lens_rows_dev.argtypes = [LP, C.c_double, C.c_double, dp, dp, ip, ll, ll, dp, dp, ip, dp, dp, vp]
    L.rtus_tt_lens_f32_rows_dev.argtypes = L.rtus_tt_lens_rows_dev.argtypes
    L.rtus_tt_layers_multi.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, C.POINTER(C.c_int), ip]
    L.rtus_tt_lens_f32_multi.argtypes = [LP, C.c_double, C.c_double, dp, dp, ip, dp, dp, ip, dp, C.POINTER(C.c_int), ip]
    pp = C.POINTER(C.c_void_p)
    L.rtus_tt_layers_multi_dev.argtypes = [dp, dp, ip, pp, pp, ip, pp, pp, ip, pp, C.POINTER(C.c_int), ip, pp, ip]
    L.rtus_tt_lens_f32_multi_dev.argtypes = [LP, C.c_double, C.c_double, pp, pp, ip, pp, pp, ip, pp, C.POINTER(C.c_int), ip, pp, ip]
    try:
        L.rtus_tt_lens_stats_dev.argtypes = [LP, C.c_double, C.c_double, dp, dp, ip, ll, ll, dp, dp, ip, dp, vp, vp]
        L.rtus_tt_lens_f32_stats_dev.argtypes = L.rtus_tt_lens_stats_dev.argtypes
        L.rtus_tt_lens_stats_dev.restype = ip
        L.rtus_tt_lens_f32_stats_dev.restype = ip
        # the planar entries with the accuracy tier as an argument (flags: RTUS_TT_TAUP_TAIL or 0)
        up = C.c_uint
        L.rtus_tt_layers_ex_dev.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, vp, up, vp]
        L.rtus_tt_layers_ex.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, vp, up, ip]
        L.rtus_tt_layers_batch_ex_dev.argtypes = [dp, dp, ip, dp, dp, ip, ll, dp, dp, ip, ll, dp, ll, ip, up, vp]
        L.rtus_tt_layers_multi_ex.argtypes = [dp, dp, ip, dp, dp, ip, dp, dp, ip, dp, C.POINTER(C.c_int), ip, up]
        L.rtus_tt_layers_multi_ex_dev.argtypes = [dp, dp, ip, pp, pp, ip, pp, pp, ip, pp, C.POINTER(C.c_int), ip, pp, ip, up]
        for name in ("rtus_tt_layers_ex_dev", "rtus_tt_layers_ex", "rtus_tt_layers_batch_ex_dev", "rtus_tt_layers_multi_ex",
                     "rtus_tt_layers_multi_ex_dev", "rtus_tt_lens_stats_dev", "rtus_tt_lens_f32_stats_dev"):
            getattr(L, name).restype = ip
    except AttributeError:                # a build from before round 4, loaded through RTUS_LIB for an A/B run: it lacks these entries
        if not os.environ.get("RTUS_LIB"):
            raise
    for name in ("rtus_tt_layers_rows_dev", "rtus_tt_lens_rows_dev", "rtus_tt_lens_f32_rows_dev", "rtus_tt_layers_multi",
                 "rtus_tt_lens_f32_multi", "rtus_tt_layers_multi_dev", "rtus_tt_lens_f32_multi_dev"):
        getattr(L, name).restype = ip
    for name in ("rtus_shoot_dev", "rtus_shoot", "rtus_match_dev", "rtus_match", "rtus_ray_hits_dev",
                 "rtus_ray_hits", "rtus_tt_layers_dev", "rtus_tt_layers", "rtus_device_count"):
        getattr(L, name).restype = ip
    _lib = L
    return L


def check(status, what):
    if status != 0:
        raise RtusError(status, what)


EXPORTS = ("rtus_strerror", "rtus_version", "rtus_last_hip_error", "rtus_device_count", "rtus_release", "rtus_selftest",
           "rtus_shoot_workspace_bytes", "rtus_shoot_dev", "rtus_shoot", "rtus_match_dev", "rtus_match",
           "rtus_ray_hits_dev", "rtus_ray_hits", "rtus_tt_layers_dev", "rtus_tt_layers", "rtus_tt_layers_batch_dev",
           "rtus_tt_lens_dev", "rtus_tt_lens", "rtus_tt_lens_f32_dev", "rtus_tt_lens_f32",
           "rtus_solve_workspace_bytes", "rtus_solve_dev", "rtus_solve",
           "rtus_focal_delays_dev", "rtus_focal_delays", "rtus_tfm_dev", "rtus_tfm",
           "rtus_tt_layers_sort_workspace_bytes", "rtus_tt_layers_sorted_dev", "rtus_table_rows_per_block", "rtus_shard_rows", "rtus_tt_layers_rows_dev", "rtus_tt_lens_rows_dev",
           "rtus_tt_lens_f32_rows_dev", "rtus_tt_layers_multi", "rtus_tt_lens_f32_multi", "rtus_tt_layers_multi_dev",
           "rtus_tt_lens_f32_multi_dev", "rtus_sweep_workspace_bytes", "rtus_sweep_dev", "rtus_sweep",
           "rtus_tt_layers_ex_dev", "rtus_tt_layers_ex", "rtus_tt_layers_batch_ex_dev", "rtus_tt_layers_multi_ex",
           "rtus_tt_layers_multi_ex_dev", "rtus_tt_lens_stats_dev", "rtus_tt_lens_f32_stats_dev")

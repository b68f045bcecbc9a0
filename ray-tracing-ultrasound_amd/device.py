"""Device-resident calls: torch tensors in HBM -> the ``*_dev`` entry points of include/rtus.h.

PyTorch is only plumbing here (device memory + the current HIP stream); every kernel is
librtus.so's.  All calls are asynchronous on ``torch.cuda.current_stream()``.
"""
import ctypes as C

import torch

from . import _lib
from .api import Params, _resolve


def _chk(t, name, dtype=torch.float64):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous CUDA tensor of {dtype}")
    return t


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t=None):
    """the current stream of the tensor's device (of the current device without a tensor)"""
    return torch.cuda.current_stream(None if t is None else t.device).cuda_stream


class ShootPlan:
    """Pre-allocated workspace + outputs for repeated forward traces of one shape (no allocation,
    no sync inside ``run`` — safe to capture in a hipGraph)."""

    def __init__(self, n_geom, n_tx, n_rays, *, want=("out8",), params: Params = None, fast=False, device="cuda"):
        self.p = _resolve(params)
        self.G, self.T, self.N = int(n_geom), int(n_tx), int(n_rays)
        self.ws_bytes = int(_lib.lib().rtus_shoot_workspace_bytes(self.N))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        G, T, N = self.G, self.T, self.N
        shapes = dict(out8=(G, T, 8, N), tof4=(G, T, 4, N), tof=(G, T, N), land_x=(G, T, N), status=(G, T, N))
        self.out = {w: torch.empty(shapes[w], dtype=torch.uint8 if w == "status" else torch.float64, device=device)
                    for w in want}
        self.lens = self.p.lens()
        self.flags = 1 if fast else 0

    def run(self, geoms, x_a, z_a, alpha, z_f, polyline_ready=False):
        """polyline_ready: the previous ``run`` of this plan used the same ``alpha`` (RTUS_POLYLINE_READY)."""
        _chk(geoms, "geoms"); _chk(x_a, "x_a"); _chk(z_a, "z_a"); _chk(alpha, "alpha"); _chk(z_f, "z_f")
        if geoms.shape != (self.G, 2) or x_a.numel() != self.T or z_a.numel() != self.T \
                or alpha.numel() != self.N or z_f.numel() != self.N:
            raise ValueError("tensor shapes do not match the plan")
        o = self.out
        st = _lib.lib().rtus_shoot_dev(C.byref(self.lens), _p(geoms), self.G, _p(x_a), _p(z_a), self.T, _p(alpha),
                                       _p(z_f), self.N, _p(o.get("out8")), _p(o.get("tof4")), _p(o.get("tof")),
                                       _p(o.get("land_x")), _p(o.get("status")), _p(self.ws), self.ws_bytes,
                                       self.flags | (8 if polyline_ready else 0), _stream())
        _lib.check(st, "rtus_shoot_dev")
        return o


class SweepPlan:
    """Pre-allocated workspace + outputs for the fused sweep (rtus_sweep_dev: forward trace + element matcher in one kernel,
    main_rt.py:464-501): first_ray i32 / hit u8 / tof_hit f64 [G, T, E], plus the per-ray arrays in ``want`` ("tof", "land_x":
    [G, T, N]).  No allocation, no sync inside ``run`` — capturable in a hipGraph."""

    def __init__(self, n_geom, n_tx, n_rays, n_rx, *, want=(), params: Params = None, fast=False, atol=1e-6, rtol=1e-5, device="cuda"):
        self.p = _resolve(params)
        self.G, self.T, self.N, self.E = int(n_geom), int(n_tx), int(n_rays), int(n_rx)
        self.ws_bytes = int(_lib.lib().rtus_sweep_workspace_bytes(self.N, self.G, self.T, self.E))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        G, T, N, E = self.G, self.T, self.N, self.E
        self.out = {"first_ray": torch.empty((G, T, E), dtype=torch.int32, device=device),
                    "hit": torch.empty((G, T, E), dtype=torch.uint8, device=device),
                    "tof_hit": torch.empty((G, T, E), dtype=torch.float64, device=device)}
        for w in want:
            if w not in ("tof", "land_x"):
                raise ValueError(f"unknown output {w!r}")
            self.out[w] = torch.empty((G, T, N), dtype=torch.float64, device=device)
        self.lens = self.p.lens()
        self.flags = 1 if fast else 0
        self.atol, self.rtol = float(atol), float(rtol)

    def run(self, geoms, x_a, z_a, alpha, z_f, x_rx, polyline_ready=False):
        """polyline_ready: the previous ``run`` of this plan used the same ``alpha`` (RTUS_POLYLINE_READY)."""
        _chk(geoms, "geoms"); _chk(x_a, "x_a"); _chk(z_a, "z_a"); _chk(alpha, "alpha"); _chk(z_f, "z_f"); _chk(x_rx, "x_rx")
        if geoms.shape != (self.G, 2) or x_a.numel() != self.T or z_a.numel() != self.T \
                or alpha.numel() != self.N or z_f.numel() != self.N or x_rx.numel() != self.E:
            raise ValueError("tensor shapes do not match the plan")
        o = self.out
        st = _lib.lib().rtus_sweep_dev(C.byref(self.lens), _p(geoms), self.G, _p(x_a), _p(z_a), self.T, _p(alpha), _p(z_f), self.N,
                                       _p(x_rx), self.E, self.atol, self.rtol, _p(o["first_ray"]), _p(o["hit"]), _p(o["tof_hit"]),
                                       _p(o.get("tof")), _p(o.get("land_x")), _p(self.ws), self.ws_bytes,
                                       self.flags | (8 if polyline_ready else 0), _stream())
        _lib.check(st, "rtus_sweep_dev")
        return o


class SolvePlan:
    """Pre-allocated workspace + outputs for repeated root-finding solves of one shape (rtus_solve_dev): no allocation,
    no sync inside ``run`` — capturable in a hipGraph.  tt / alpha_root [G, T, E]; with all_roots also tt_all / alpha_all
    [G, T, E, 4] and n_roots [G, T, E]."""

    def __init__(self, n_geom, n_tx, n_rays, n_rx, *, params: Params = None, fast=False, true_tangent=False,
                 analytic_lens=False, all_roots=False, device="cuda", one_lane=False, three_launches=False):
        self.p = _resolve(params)
        self.G, self.T, self.N, self.E = int(n_geom), int(n_tx), int(n_rays), int(n_rx)
        self.ws_bytes = int(_lib.lib().rtus_solve_workspace_bytes(self.N, self.G, self.T, self.E))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        G, T, E = self.G, self.T, self.E
        f64 = dict(dtype=torch.float64, device=device)
        self.out = {"tt": torch.empty((G, T, E), **f64), "alpha_root": torch.empty((G, T, E), **f64)}
        if all_roots:
            self.out.update(tt_all=torch.empty((G, T, E, 4), **f64), alpha_all=torch.empty((G, T, E, 4), **f64),
                            n_roots=torch.empty((G, T, E), dtype=torch.uint8, device=device))
        self.lens = self.p.lens()
        self.flags = (1 if fast else 0) | (2 if true_tangent else 0) | (4 if analytic_lens else 0) | (0x10 if one_lane else 0) | (0x20 if three_launches else 0)

    def run(self, geoms, x_a, z_a, alpha, x_rx, z_land=None, polyline_ready=False):
        """polyline_ready: the previous ``run`` of this plan used the same ``alpha`` tensor contents (RTUS_POLYLINE_READY: the
        lens polyline in the workspace is kept instead of rebuilt)."""
        _chk(geoms, "geoms"); _chk(x_a, "x_a"); _chk(z_a, "z_a"); _chk(alpha, "alpha"); _chk(x_rx, "x_rx")
        if geoms.shape != (self.G, 2) or x_a.numel() != self.T or z_a.numel() != self.T \
                or alpha.numel() != self.N or x_rx.numel() != self.E:
            raise ValueError("tensor shapes do not match the plan")
        o = self.out
        st = _lib.lib().rtus_solve_dev(C.byref(self.lens), _p(geoms), self.G, _p(x_a), _p(z_a), self.T, _p(alpha), self.N,
                                       _p(x_rx), self.E, float(self.p.d if z_land is None else z_land), _p(o["tt"]),
                                       _p(o["alpha_root"]), _p(o.get("tt_all")), _p(o.get("alpha_all")), _p(o.get("n_roots")),
                                       _p(self.ws), self.ws_bytes, self.flags | (8 if polyline_ready else 0), _stream())
        _lib.check(st, "rtus_solve_dev")
        return o


def match_dev(land_x, tof, x_rx, atol=1e-6, rtol=1e-5, out=None):
    """land_x/tof [rows, N], x_rx [E] -> (first_ray i32[rows,E], hit u8[rows,E], tof_hit f64[rows,E])."""
    _chk(land_x, "land_x"); _chk(tof, "tof"); _chk(x_rx, "x_rx")
    rows, N, E = land_x.shape[0], land_x.shape[1], x_rx.numel()
    if out is None:
        out = (torch.empty((rows, E), dtype=torch.int32, device=land_x.device),
               torch.empty((rows, E), dtype=torch.uint8, device=land_x.device),
               torch.empty((rows, E), dtype=torch.float64, device=land_x.device))
    first, hit, tof_hit = out
    st = _lib.lib().rtus_match_dev(_p(land_x), _p(tof), rows, N, _p(x_rx), E, float(atol), float(rtol), _p(first),
                                   _p(hit), _p(tof_hit), _stream())
    _lib.check(st, "rtus_match_dev")
    return out


TAUP_TAIL = 0x1             # RTUS_TT_TAUP_TAIL: the faster accuracy tier of the planar solver (include/rtus.h)


def tt_layers_dev(z_if, c, xe, ze, xf, zf, out=None, iters=None, row0=0, n_rows_total=None, taup=False):
    """Fermat travel times through horizontal layers; z_if/c are small HOST sequences.

    row0 / n_rows_total: xe, ze are rows [row0, row0 + len(xe)) of a table of n_rows_total rows (rtus_tt_layers_rows_dev): with
    row0 a multiple of ``rows_per_block(n_rows_total, n_f)`` the block comes out with the bits the whole table's launch gives it."""
    import numpy as np
    z_if = np.ascontiguousarray(z_if, dtype=np.float64).reshape(-1)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(-1)
    if c.size != z_if.size + 1:               # the C layer reads c[0 .. n_if] from this host pointer
        raise ValueError("need len(c) == len(z_if) + 1")
    _chk(xe, "xe"); _chk(ze, "ze"); _chk(xf, "xf"); _chk(zf, "zf")
    n_e, n_f = xe.numel(), xf.numel()
    if ze.numel() != n_e or zf.numel() != n_f:
        raise ValueError("xe/ze and xf/zf must pair up")
    if out is None:
        out = torch.empty((n_e, n_f), dtype=torch.float64, device=xe.device)
    _chk(out, "out")
    if out.numel() != n_e * n_f:
        raise ValueError("out has the wrong size")
    if iters is not None and (n_rows_total is not None or taup):
        raise ValueError("iters is a whole-table diagnostic of the accurate tier")
    if n_rows_total is not None:
        st = _lib.lib().rtus_tt_layers_rows_dev(z_if.ctypes.data if z_if.size else None, c.ctypes.data, z_if.size, _p(xe), _p(ze),
                                                n_e, int(row0), int(n_e if n_rows_total is None else n_rows_total), _p(xf), _p(zf), n_f,
                                                _p(out), TAUP_TAIL if taup else 0, _stream())
        _lib.check(st, "rtus_tt_layers_rows_dev")
        return out
    st = _lib.lib().rtus_tt_layers_ex_dev(z_if.ctypes.data if z_if.size else None, c.ctypes.data, z_if.size, _p(xe),
                                          _p(ze), n_e, _p(xf), _p(zf), n_f, _p(out), _p(iters), TAUP_TAIL if taup else 0, _stream())
    _lib.check(st, "rtus_tt_layers_ex_dev")
    return out


def tt_layers_sorted_dev(z_if, c, xe, ze, xf, zf, out=None, ws=None, taup=False):
    """The planar table for an aperture handed over in ANY order (rtus_tt_layers_sorted_dev): sorted by (depth, position) on the
    device, every row stored where it belongs.  ``ws``: optional uint8 workspace tensor to reuse between calls."""
    import numpy as np
    z_if = np.ascontiguousarray(z_if, dtype=np.float64).reshape(-1)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(-1)
    if c.size != z_if.size + 1:
        raise ValueError("need len(c) == len(z_if) + 1")
    _chk(xe, "xe"); _chk(ze, "ze"); _chk(xf, "xf"); _chk(zf, "zf")
    n_e, n_f = xe.numel(), xf.numel()
    if ze.numel() != n_e or zf.numel() != n_f:
        raise ValueError("xe/ze and xf/zf must pair up")
    if out is None:
        out = torch.empty((n_e, n_f), dtype=torch.float64, device=xe.device)
    _chk(out, "out")
    need = int(_lib.lib().rtus_tt_layers_sort_workspace_bytes(n_e))
    if ws is None:
        ws = torch.empty(need, dtype=torch.uint8, device=xe.device)
    if ws.numel() < need or out.numel() != n_e * n_f:
        raise ValueError("workspace or out too small")
    st = _lib.lib().rtus_tt_layers_sorted_dev(z_if.ctypes.data if z_if.size else None, c.ctypes.data, z_if.size, _p(xe), _p(ze), n_e,
                                              _p(xf), _p(zf), n_f, _p(out), _p(ws), ws.numel(), TAUP_TAIL if taup else 0, _stream(xe))
    _lib.check(st, "rtus_tt_layers_sorted_dev")
    return out


def rows_per_block(n_rows_total, n_f, dtype=torch.float64):
    """Rows the table kernels solve per workgroup for a table of this size: shard boundaries that are multiples of it
    reproduce the one-launch table bit for bit (rtus_table_rows_per_block)."""
    r = _lib.lib().rtus_table_rows_per_block(int(n_rows_total), int(n_f), 8 if dtype == torch.float64 else 4)
    if r < 0:
        _lib.check(r, "rtus_table_rows_per_block")
    return int(r)


def tt_lens_rows_dev(xe, ze, xf, zf, out, *, params: Params = None, alpha_lo=None, alpha_hi=None, row0=0, n_rows_total=None):
    """Curved-lens table rows [row0, row0 + len(xe)) of an n_rows_total-row table, fp64 or fp32 by the tensors' dtype."""
    from .api import ALPHA_MAX
    p = _resolve(params)
    f64 = xe.dtype == torch.float64
    for t, n in ((xe, "xe"), (ze, "ze"), (xf, "xf"), (zf, "zf"), (out, "out")):
        _chk(t, n, xe.dtype)
    n_e, n_f = xe.numel(), xf.numel()
    if ze.numel() != n_e or zf.numel() != n_f or out.numel() != n_e * n_f:
        raise ValueError("xe/ze, xf/zf and out must pair up")
    fn = _lib.lib().rtus_tt_lens_rows_dev if f64 else _lib.lib().rtus_tt_lens_f32_rows_dev
    lens = p.lens()
    st = fn(C.byref(lens), -ALPHA_MAX if alpha_lo is None else float(alpha_lo), ALPHA_MAX if alpha_hi is None else float(alpha_hi),
            _p(xe), _p(ze), n_e, int(row0), int(n_e if n_rows_total is None else n_rows_total), _p(xf), _p(zf), n_f, _p(out), None,
            _stream())
    _lib.check(st, "rtus_tt_lens_rows_dev")
    return out


def tt_lens_stats_dev(xe, ze, xf, zf, out, *, params: Params = None, alpha_lo=None, alpha_hi=None, row0=0, n_rows_total=None):
    """tt_lens_rows_dev + how the rows were solved (rtus_tt_lens[_f32]_stats_dev) -> (out, dict of wave-element counts)."""
    from .api import ALPHA_MAX
    p = _resolve(params)
    f64 = xe.dtype == torch.float64
    for t, n in ((xe, "xe"), (ze, "ze"), (xf, "xf"), (zf, "zf"), (out, "out")):
        _chk(t, n, xe.dtype)
    n_e, n_f = xe.numel(), xf.numel()
    if ze.numel() != n_e or zf.numel() != n_f or out.numel() != n_e * n_f:
        raise ValueError("xe/ze, xf/zf and out must pair up")
    stats = torch.zeros(5, dtype=torch.int64, device=xe.device)
    fn = _lib.lib().rtus_tt_lens_stats_dev if f64 else _lib.lib().rtus_tt_lens_f32_stats_dev
    lens = p.lens()
    st = fn(C.byref(lens), -ALPHA_MAX if alpha_lo is None else float(alpha_lo), ALPHA_MAX if alpha_hi is None else float(alpha_hi),
            _p(xe), _p(ze), n_e, int(row0), int(n_e if n_rows_total is None else n_rows_total), _p(xf), _p(zf), n_f, _p(out), _p(stats),
            _stream())
    _lib.check(st, "rtus_tt_lens_stats_dev")
    v = [int(x) for x in stats.cpu()]
    return out, dict(t_only=v[0], one_evaluation=v[1], iterated=v[2], scanned=v[3], iteration_evaluations=v[4],
                     wave_elements=n_e * 4 * ((n_f + 255) // 256))     # waves launched per row: whole workgroups of 256 targets


def tt_layers_batch_dev(z_if, c, xe, ze, xf, zf, out=None, taup=False):
    """B independent problems of one shape and one medium in ONE launch (rtus_tt_layers_batch_ex_dev; taup: the faster accuracy tier).

    xe/ze: [B, n_e] or [n_e] (one aperture shared by all problems); xf/zf: [B, n_f] or [n_f] (shared) -> tt [B, n_e, n_f].
    At least one of the two must carry the batch dimension."""
    import numpy as np
    z_if = np.ascontiguousarray(z_if, dtype=np.float64).reshape(-1)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(-1)
    if c.size != z_if.size + 1:
        raise ValueError("need len(c) == len(z_if) + 1")
    for t, n in ((xe, "xe"), (ze, "ze"), (xf, "xf"), (zf, "zf")):
        _chk(t, n)
    if xe.shape != ze.shape or xf.shape != zf.shape or xe.dim() not in (1, 2) or xf.dim() not in (1, 2):
        raise ValueError("xe/ze and xf/zf must pair up, as [B, n] or [n]")
    B = xe.shape[0] if xe.dim() == 2 else (xf.shape[0] if xf.dim() == 2 else None)
    if B is None or (xe.dim() == 2 and xf.dim() == 2 and xe.shape[0] != xf.shape[0]):
        raise ValueError("need a batch dimension on xe/ze and / or xf/zf (equal when on both)")
    n_e, n_f = xe.shape[-1], xf.shape[-1]
    if out is None:
        out = torch.empty((B, n_e, n_f), dtype=torch.float64, device=xe.device)
    _chk(out, "out")
    if out.numel() != B * n_e * n_f:
        raise ValueError("out has the wrong size")
    st = _lib.lib().rtus_tt_layers_batch_ex_dev(z_if.ctypes.data if z_if.size else None, c.ctypes.data, z_if.size, _p(xe), _p(ze),
                                                n_e, n_e if xe.dim() == 2 else 0, _p(xf), _p(zf), n_f,
                                                n_f if xf.dim() == 2 else 0, _p(out), n_e * n_f, B, TAUP_TAIL if taup else 0, _stream())
    _lib.check(st, "rtus_tt_layers_batch_ex_dev")
    return out


def focal_delays_dev(tt, out=None):
    """delays[e, f] = max_e' tt[e', f] - tt[e, f] on device (NaN-aware); ``out`` may be ``tt`` itself."""
    _chk(tt, "tt")
    if tt.dim() != 2:
        raise ValueError("tt must be [n_elem, n_focal]")
    if out is None:
        out = torch.empty_like(tt)
    _chk(out, "out")
    if out.shape != tt.shape:
        raise ValueError("out has the wrong shape")
    st = _lib.lib().rtus_focal_delays_dev(_p(tt), tt.shape[0], tt.shape[1], _p(out), _stream())
    _lib.check(st, "rtus_focal_delays_dev")
    return out


def tfm_dev(fmc, fs, tt_tx, tt_rx=None, t0=0.0, out=None):
    """TFM delay-and-sum on device: fmc float32 [n_tx, n_rx, n_t], tt_tx [n_tx, n_f], tt_rx [n_rx, n_f] (default: tt_tx)
    -> image float32 [n_f].  tt_* may be row blocks / column slices of a larger table as long as they are contiguous."""
    _chk(fmc, "fmc", torch.float32); _chk(tt_tx, "tt_tx")
    tt_rx = tt_tx if tt_rx is None else _chk(tt_rx, "tt_rx")
    if fmc.dim() != 3 or tt_tx.dim() != 2 or tt_rx.dim() != 2 or tt_tx.shape[1] != tt_rx.shape[1] \
            or fmc.shape[0] != tt_tx.shape[0] or fmc.shape[1] != tt_rx.shape[0]:
        raise ValueError("need fmc [n_tx, n_rx, n_t], tt_tx [n_tx, n_f], tt_rx [n_rx, n_f]")
    n_f = tt_tx.shape[1]
    if out is None:
        out = torch.empty(n_f, dtype=torch.float32, device=fmc.device)
    _chk(out, "out", torch.float32)
    if out.numel() != n_f or not (out.device == fmc.device == tt_tx.device == tt_rx.device):
        raise ValueError("out must hold n_focal float32 values on the device of fmc / tt_tx / tt_rx")
    st = _lib.lib().rtus_tfm_dev(_p(fmc), fmc.shape[0], fmc.shape[1], fmc.shape[2], float(fs), float(t0), _p(tt_tx), _p(tt_rx), n_f,
                                 _p(out), _stream(fmc))
    _lib.check(st, "rtus_tfm_dev")
    return out


class LayersPlan:
    """Pre-bound ``rtus_tt_layers_dev`` call for repeated solves of one shape: ``run()`` is a single
    ctypes call (no argument checking, no allocation, no sync) — capturable in a hipGraph."""

    def __init__(self, z_if, c, xe, ze, xf, zf, out=None, iters=None, row0=0, n_rows_total=None, taup=False):
        import numpy as np
        self.z_if = np.ascontiguousarray(z_if, dtype=np.float64).reshape(-1)
        self.c = np.ascontiguousarray(c, dtype=np.float64).reshape(-1)
        if self.c.size != self.z_if.size + 1:
            raise ValueError("need len(c) == len(z_if) + 1")
        for t, n in ((xe, "xe"), (ze, "ze"), (xf, "xf"), (zf, "zf")):
            _chk(t, n)
        self.n_e, self.n_f = xe.numel(), xf.numel()
        if ze.numel() != self.n_e or zf.numel() != self.n_f:
            raise ValueError("xe/ze and xf/zf must pair up")
        self.out = out if out is not None else torch.empty((self.n_e, self.n_f), dtype=torch.float64, device=xe.device)
        _chk(self.out, "out")
        if self.out.numel() != self.n_e * self.n_f:
            raise ValueError("out has the wrong size")
        self._keep = (xe, ze, xf, zf, self.out, iters)
        self._fn = _lib.lib().rtus_tt_layers_dev
        self._args = [self.z_if.ctypes.data if self.z_if.size else None, self.c.ctypes.data, self.z_if.size,
                      _p(xe), _p(ze), self.n_e, _p(xf), _p(zf), self.n_f, _p(self.out), _p(iters)]
        if n_rows_total is not None or taup:      # a row block of a larger table and / or the tau-p tier (rtus_tt_layers_rows_dev)
            self._fn = _lib.lib().rtus_tt_layers_rows_dev
            self._args = self._args[:6] + [int(row0), int(self.n_e if n_rows_total is None else n_rows_total)] + self._args[6:10] + \
                [TAUP_TAIL if taup else 0]

    def run(self, stream=None):
        st = self._fn(*self._args, _stream() if stream is None else stream)
        if st:
            _lib.check(st, "rtus_tt_layers_dev")
        return self.out

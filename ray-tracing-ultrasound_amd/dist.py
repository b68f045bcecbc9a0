"""Multi-GPU sharding of the (tx element x target) travel-time matrix — one process per GPU.

Every (element, target) solve is independent, so the matrix shards by contiguous tx-row blocks with
no collective on the compute path.  The single exchange is the reassembly of the row blocks on every
rank: an in-place ``all_gather_into_tensor`` (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the
CPU tests).  Rows are padded to a multiple of the world size so all shards have equal size
(SURVEY 8(e)); the pad rows are trimmed from the returned matrix.
"""
import torch
import torch.distributed as dist


def row_shard(n_rows: int, world: int, rank: int, align: int = 1):
    """(lo, hi, rows_per_rank): rank owns rows [lo, hi) of the padded world*rows_per_rank layout.

    align: rows_per_rank is rounded up to a multiple of it.  With align = the table kernels' rows per workgroup
    (device.rows_per_block) every shard starts on a block boundary of the whole table, and the sharded table is bit for bit
    the one-GPU table (a row is always solved in the same workgroup with the same predecessors)."""
    if n_rows <= 0 or world <= 0 or not (0 <= rank < world) or align <= 0:
        raise ValueError("bad shard request")
    per = -(-n_rows // world)
    per = -(-per // align) * align
    lo = min(rank * per, n_rows)
    hi = min(lo + per, n_rows)
    return lo, hi, per


class RowShardedMatrix:
    """Double-buffered full matrix [world*per, n_cols]; each rank's kernel writes its row block into
    ``local(slot)`` and ``gather(slot)`` reassembles all blocks on every rank.  With more than one rank the
    block is its own buffer (send and receive buffers of the collective never alias); with one rank it is a
    view of the matrix and no collective runs."""

    def __init__(self, n_rows, n_cols, *, dtype=torch.float64, device="cuda", slots=2, group=None, align=1):
        self.group = group
        self.align = int(align)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.lo, self.hi, self.per = row_shard(self.n_rows, self.world, self.rank, self.align)
        self.full = [torch.zeros((self.world * self.per, self.n_cols), dtype=dtype, device=device)
                     for _ in range(slots)]
        self.shard = ([f[:self.per] for f in self.full] if self.world == 1 else
                      [torch.zeros((self.per, self.n_cols), dtype=dtype, device=device) for _ in range(slots)])
        self._work = [None] * slots

    def local(self, slot=0):
        """The padded [per, n_cols] block this rank computes (rows beyond hi-lo are padding)."""
        return self.shard[slot]

    def local_block(self, slot=0):
        """(lo, hi, rows) — this rank's rows [lo, hi) of the table as a [hi-lo, n_cols] view, WITHOUT any exchange.
        A consumer that works row by row (per-tx focal laws, a per-tx partial sum) takes this and skips ``gather``:
        reassembling the table costs every GPU 7x its own shard over xGMI — an order of magnitude more than computing it
        (DESIGN.md section 6)."""
        return self.lo, self.hi, self.shard[slot][:self.hi - self.lo]

    def wait(self, slot=0):
        w = self._work[slot]
        if w is not None:
            w.wait()
            self._work[slot] = None

    def gather(self, slot=0, async_op=False):
        """All-gather of the row blocks into the full matrix.  With async_op the collective overlaps later work
        on the current stream; call ``wait(slot)`` before reading or rewriting the slot."""
        if self.world == 1:
            return None
        self.wait(slot)
        w = dist.all_gather_into_tensor(self.full[slot], self.local(slot), group=self.group, async_op=async_op)
        self._work[slot] = w if async_op else None
        return w

    def matrix(self, slot=0):
        """The reassembled [n_rows, n_cols] matrix (pad rows trimmed)."""
        self.wait(slot)
        return self.full[slot][:self.n_rows]


def sharded_rows(n_rows, n_cols, row_solver, *, dtype=torch.float64, device="cuda", group=None, align=1):
    """Generic row-sharded table: ``row_solver(lo, hi, out)`` fills rows [lo, hi) of the table into ``out``
    ([hi-lo, n_cols] view of this rank's block); the blocks are then all-gathered in place on every rank."""
    m = RowShardedMatrix(n_rows, n_cols, dtype=dtype, device=device, slots=1, group=group, align=align)
    n_own = m.hi - m.lo
    if n_own > 0:
        row_solver(m.lo, m.hi, m.local(0)[:n_own])
    m.gather(0)
    return m.matrix(0)


def travel_time_layers_sharded(z_if, c, xe, ze, xf, zf, *, group=None, solver=None, align=None):
    """Element x focal travel-time matrix computed by row shards on all ranks, gathered everywhere — bit for bit the one-GPU
    table: shards start on the table's workgroup-block boundaries and the kernel is told where its rows sit in the table.

    xe, ze, xf, zf: torch tensors on this rank's device holding the FULL element / focal lists.
    ``solver(z_if, c, xe_shard, ze_shard, xf, zf, out=, row0=, n_rows_total=)`` defaults to the HIP kernel; the CPU (gloo)
    tests inject the oracle here (with ``align`` given) — the product default never touches it.
    """
    n_rows = xe.numel()
    if solver is None:
        from .device import rows_per_block, tt_layers_dev as solver
        align = rows_per_block(n_rows, xf.numel()) if align is None else align

    def rows(lo, hi, out):
        xs, zs = xe[lo:hi].contiguous(), ze[lo:hi].contiguous()
        solver(z_if, c, xs, zs, xf, zf, out=out, row0=lo, n_rows_total=n_rows)
        if xs.is_cuda:
            torch.cuda.current_stream(xs.device).synchronize()     # xs / zs (possibly temporaries) must outlive the asynchronous launch
    return sharded_rows(n_rows, xf.numel(), rows, dtype=xe.dtype, device=xe.device, group=group, align=1 if align is None else align)


def travel_time_lens_sharded(xe, ze, xf, zf, *, params=None, alpha_lo=None, alpha_hi=None, group=None):
    """BASELINE config 4 across ranks: curved-lens Fermat table (fp64 or fp32 by the tensors' dtype), tx rows
    sharded on the table's workgroup-block boundaries (bit for bit the one-GPU table), reassembled by the all-gather."""
    from .device import rows_per_block, tt_lens_rows_dev
    n_rows = xe.numel()

    def rows(lo, hi, out):
        xs, zs = xe[lo:hi].contiguous(), ze[lo:hi].contiguous()
        tt_lens_rows_dev(xs, zs, xf, zf, out, params=params, alpha_lo=alpha_lo, alpha_hi=alpha_hi, row0=lo, n_rows_total=n_rows)
        torch.cuda.current_stream().synchronize()          # xs / zs must outlive the launch
    return sharded_rows(n_rows, xf.numel(), rows, dtype=xe.dtype, device=xe.device, group=group,
                        align=rows_per_block(n_rows, xf.numel(), xe.dtype))


def col_shard(n_cols: int, world: int, rank: int):
    """(lo, hi, cols_per_rank): rank owns columns (focal points) [lo, hi) of the padded world*cols_per_rank layout."""
    return row_shard(n_cols, world, rank)


def image_sharded(n_focal, slice_solver, *, dtype=torch.float32, device="cuda", group=None):
    """Focal-point (column) sharding for consumers that need EVERY element's travel time per focal point — the TFM
    delay-and-sum does: image[f] sums over all (tx, rx).  Sharding the table's rows would force the all-gather of the
    whole table first; sharding its COLUMNS needs no exchange of travel times at all: rank r solves the table for all
    elements on its slice of focal points, beamforms the slice, and only the image slices (4 B per focal point) are
    all-gathered.  ``slice_solver(lo, hi, out)`` fills ``out`` ([hi-lo]) with the image of focal points [lo, hi)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi, per = col_shard(n_focal, world, rank)
    mine = torch.zeros(per, dtype=dtype, device=device)
    if hi > lo:
        slice_solver(lo, hi, mine[:hi - lo])
    if world == 1:
        return mine[:n_focal]
    full = torch.empty(world * per, dtype=dtype, device=device)
    dist.all_gather_into_tensor(full, mine, group=group)
    return full[:n_focal]


def tfm_layers_sharded(fmc, fs, z_if, c, xe, ze, xf, zf, *, t0=0.0, group=None, table=None, beamform=None):
    """TFM image of a layered medium across ranks with NO travel-time exchange (see :func:`image_sharded`): each rank
    solves tt[all elements, its focal slice] with the HIP Fermat kernel and beamforms it with the HIP TFM kernel.
    fmc [n_el, n_el, n_t] float32 and the element / focal coordinate tensors are replicated on every rank.
    ``table`` / ``beamform`` default to the HIP kernels; the CPU (gloo) tests inject oracle stand-ins."""
    if table is None:
        from .device import tt_layers_dev as table
    if beamform is None:
        from .device import tfm_dev as beamform

    def one_slice(lo, hi, out):
        tt = table(z_if, c, xe, ze, xf[lo:hi].contiguous(), zf[lo:hi].contiguous())
        beamform(fmc, fs, tt, None, t0, out)
    return image_sharded(xf.numel(), one_slice, dtype=torch.float32, device=xf.device, group=group)

"""Multi-GPU sharding of the hot path: one process per GPU, no data-path collective.

Fusion shards by output: every (timepoint, region, channel, z) plane is independent
(reference precedent: per-FOV independent writes, zarr_stitcher.py:443-489), so planes (or whole
regions) are dealt to ranks and never exchanged.  Registration shards by pair.  The only
collective is an all-gather of the small shift table (RCCL over xGMI on the GPU box, gloo in
the CPU tests): a few int32 per region, latency-bound.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .placement import Shifts

SHIFT_ROW = 8   # valid, h_dy, h_dx, v_dy, v_dx, rev_dy, rev_dx, rev_odd  (rev_* = 0 and valid&2 unset without S-Pattern)


def rank_and_world(group=None) -> Tuple[int, int]:
    """(rank, world_size) of the default process group, (0, 1) when not distributed."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def collective_device(stitcher):
    """Where collective tensors must live: the GPU for nccl (RCCL), the host for gloo."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
        return stitcher.device
    return None


def broadcast_object(obj, src: int = 0):
    """Small host objects every rank must agree on (the timestamped output folder, an estimated
    flatfield): rank ``src``'s value everywhere.  Identity when not distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return obj
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def barrier() -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def block_cyclic(n_items: int, rank: int, world: int) -> List[int]:
    """Indices of the work items rank ``rank`` owns: i with i % world == rank."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_items, world))


def contiguous_blocks(n_items: int, rank: int, world: int) -> List[int]:
    """Indices of the work items rank ``rank`` owns when every rank takes ONE contiguous run (sizes differ by at most
    one, the longer runs first).  Used for the (channel, z) planes of a region: consecutive planes are the z planes of
    one channel, which share a gain image -- kept on one rank they go through the fusion kernel together (plane groups,
    csrc/fuse.hip) and their files sit side by side."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def row_bands(canvas_h: int, num_levels: int, chunk_rows: int = 512) -> List[Tuple[int, int]]:
    """Level-0 row bands [y0, y1) that ONE (channel, z) plane can be cut into so that several ranks fuse and write it
    at once (the finest grain; precedent: the reference's per-FOV writers into one pre-created array,
    zarr_stitcher.py:395-440).  A band is ``chunk_rows * 2^(num_levels - 1)`` rows: it then starts on a chunk-row
    boundary of EVERY pyramid level (level l halves the rows l times, Scaler.nearest: row y of level l+1 is row 2y+1 of
    level l), so a rank computes and writes all levels of its bands without touching another rank's chunks.  A canvas
    shorter than one such band is a single band."""
    band = int(chunk_rows) << max(0, int(num_levels) - 1)
    return [(y0, min(y0 + band, int(canvas_h))) for y0 in range(0, max(1, int(canvas_h)), band)]


def plane_band_units(n_planes: int, bands: Sequence[Tuple[int, int]], rank: int, world: int) -> List[Tuple[int, int]]:
    """(plane, band index) work units of rank ``rank``: planes first when there are enough of them (whole planes,
    no tile is read twice; every rank a contiguous run of planes), else every plane cut into its bands, dealt
    block-cyclically."""
    if n_planes >= world or len(bands) <= 1:
        return [(p, -1) for p in contiguous_blocks(n_planes, rank, world)]      # -1: the whole plane
    units = [(p, b) for p in range(n_planes) for b in range(len(bands))]
    return [units[i] for i in block_cyclic(len(units), rank, world)]


def shifts_to_row(s: Optional[Shifts]) -> np.ndarray:
    row = np.zeros(SHIFT_ROW, dtype=np.int32)
    if s is None:
        return row
    row[0] = 1 | (2 if s.h_shift_rev is not None else 0)
    row[1:3] = s.h_shift
    row[3:5] = s.v_shift
    if s.h_shift_rev is not None:
        row[5:7] = s.h_shift_rev
        row[7] = int(s.h_shift_rev_odd)
    return row


def row_to_shifts(row: Sequence[int]) -> Optional[Shifts]:
    row = [int(v) for v in row]
    if not row[0] & 1:
        return None
    rev = (row[5], row[6]) if row[0] & 2 else None
    return Shifts((row[1], row[2]), (row[3], row[4]), rev, row[7])


class PendingShiftTable:
    """An all-gather of shift rows in flight; ``result()`` waits for it and returns the table."""

    def __init__(self, table=None, work=None, out=None, keep=None):
        self._table, self._work, self._out, self._keep = table, work, out, keep

    def result(self) -> np.ndarray:
        if self._table is None:
            self._work.wait()
            self._table = self._out.cpu().numpy().reshape(-1, SHIFT_ROW)
            self._work = self._out = self._keep = None
        return self._table


def all_gather_shift_table_async(local_rows: np.ndarray, device=None, group=None) -> PendingShiftTable:
    """Start the all-gather of ``local_rows`` [k, SHIFT_ROW] int32 (same k on every rank) and return at
    once.  Nothing on a rank's own data path needs the other ranks' rows (every rank fuses with the
    shifts of its own regions), so the collective runs beside the fusion launch on RCCL's stream and
    is only waited for when the table is written out."""
    import torch
    import torch.distributed as dist
    local_rows = np.ascontiguousarray(local_rows, dtype=np.int32).reshape(-1, SHIFT_ROW)
    import os
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size(group) == 1 and not os.environ.get('SQ_DIST_FORCE_COLLECTIVE')):
        return PendingShiftTable(table=local_rows.copy())
    world = dist.get_world_size(group)
    t = torch.from_numpy(local_rows)
    if device is not None:
        t = t.to(device)
    out = torch.empty((world * t.shape[0], SHIFT_ROW), dtype=t.dtype, device=t.device)
    work = dist.all_gather_into_tensor(out, t, group=group, async_op=True)
    return PendingShiftTable(work=work, out=out, keep=t)


def all_gather_shift_table(local_rows: np.ndarray, device=None, group=None) -> np.ndarray:
    """Every rank contributes ``local_rows`` [k, SHIFT_ROW] int32 (same k on every rank) and gets the
    table [world * k, SHIFT_ROW] back, rank-major.  Single process: returns the input."""
    return all_gather_shift_table_async(local_rows, device, group).result()


PAIR_ROW = 3    # dy, dx, err (float64) of one registered pair (all-pairs mode)


def all_gather_pair_table(local_rows: np.ndarray, n_pairs: int, rank: int, world: int, device=None, group=None) -> np.ndarray:
    """The all-pairs shift table: rank ``r`` registered the contiguous run ``contiguous_blocks(n_pairs, r, world)`` of
    the pair list and contributes its [k_r, 3] float64 rows {dy, dx, err}; every rank gets the whole [n_pairs, 3] table
    back in pair order.  Runs differ in length by at most one, so the rows travel in equal slots of ceil(n_pairs /
    world) rows (the spare one NaN) through ONE all_gather_into_tensor -- 32x32 grid: 1984 pairs x 24 B = 47 KB,
    latency-bound over xGMI.  ``world == 1`` (a single process, or one rank of a
    larger run registering a region on its own): the input, no collective."""
    import torch
    import torch.distributed as dist
    import os
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64).reshape(-1, PAIR_ROW)
    mine = contiguous_blocks(n_pairs, rank, world)
    if len(local_rows) != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} of {n_pairs} pairs but brought {len(local_rows)} rows")
    # world == 1 means "this caller registered every pair itself": no collective, whatever process group exists around it
    # (a rank of an N-rank run that registers one of ITS OWN regions -- per-region registration with the regions dealt to
    # the ranks -- is such a caller; comparing its world of 1 with the group's N used to raise on every rank)
    if world == 1 and not os.environ.get('SQ_DIST_FORCE_COLLECTIVE'):
        return local_rows.copy()
    if not (dist.is_available() and dist.is_initialized()):
        if world != 1:
            raise RuntimeError(f"world of {world} ranks but no process group: nothing to gather the pair table over")
        return local_rows.copy()
    if dist.get_world_size(group) != world:
        raise ValueError(f"world {world} != process group size {dist.get_world_size(group)}")
    slot = -(-n_pairs // world) if n_pairs else 0
    if slot == 0:
        return np.zeros((0, PAIR_ROW), dtype=np.float64)
    padded = np.full((slot, PAIR_ROW), np.nan, dtype=np.float64)
    padded[:len(local_rows)] = local_rows
    t = torch.from_numpy(padded)
    if device is not None:
        t = t.to(device)
    out = torch.empty((world * slot, PAIR_ROW), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    gathered = out.cpu().numpy().reshape(world, slot, PAIR_ROW)
    table = np.empty((n_pairs, PAIR_ROW), dtype=np.float64)
    for r in range(world):
        idx = contiguous_blocks(n_pairs, r, world)
        table[idx] = gathered[r, :len(idx)]
    return table


def first_valid(table: np.ndarray) -> Optional[Shifts]:
    """The reference registers once and applies the result everywhere (stitcher.py:1244-1246):
    the first valid row of the gathered table is that result."""
    for row in table:
        s = row_to_shifts(row)
        if s is not None:
            return s
    return None

"""Host side of registration: builds crop pairs, calls the HIP pipeline through the C-ABI and
finishes the few scalar steps the reference does in Python/numpy float64.

Replaces skimage.registration.phase_cross_correlation as called from
calculate_horizontal_shift / calculate_vertical_shift (stitcher.py:500-524) and the pair
selection of calculate_shifts (stitcher.py:455-496).  ``register_all_pairs`` is the
north-star extension (every adjacent pair, batched); the reference itself registers only the
centre pairs.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import native, placement
from .placement import Shifts

def crop_length_supported(n: int) -> bool:
    """Whether the device pipeline takes a crop side of ``n`` pixels (include/squidstitch.h: 2 ... 65535; a line that fits
    the 160 KB of LDS -- any length up to 4860, smooth lengths up to 9720 -- is transformed there, a longer one in the
    workspace)."""
    return bool(native.lib().sq_register_line_supported(int(n)))


def check_crop_lengths(height: int, width: int, max_x_overlap: int, max_y_overlap: int) -> None:
    """Raise before any work if the registration crops of a ``height x width`` tile have a side the device pipeline
    does not take.  The reference (pocketfft) has no such limit; this one is a crop side of 65535 pixels."""
    sides = []
    for make, ov in ((placement.horizontal_crop_origins, max_x_overlap), (placement.vertical_crop_origins, max_y_overlap)):
        try:
            sides.extend(make(height, width, int(ov))[:2])
        except ValueError:      # a direction with nothing to register (one row / one column): reported where it is used
            pass
    bad = sorted({int(v) for v in sides if not crop_length_supported(v)})
    if bad:
        raise ValueError(f"registration crops of a {height} x {width} tile have sides of {bad} pixels; the device pipeline takes "
                         "crop sides of 2 ... 65535 pixels. Stitch without -r, or supply h_shift / v_shift.")


NORMALIZATIONS = {'phase': native.SQ_NORM_PHASE, None: native.SQ_NORM_NONE, 'none': native.SQ_NORM_NONE}


def _norm_code(normalization) -> int:
    if normalization not in NORMALIZATIONS:
        raise ValueError("normalization must be either phase or None")   # skimage's message
    return NORMALIZATIONS[normalization]


def shifts_from_results(res: np.ndarray, upsample_factor: int):
    """RESULT_DTYPE rows -> (shifts [n,2] float64, error [n], phasediff [n]) with the float64
    arithmetic of skimage (_phase_cross_correlation.py:232-250, :91-106, :78-88)."""
    shifts = res['coarse'].astype(np.float64)
    if upsample_factor > 1:
        shifts = np.round(shifts * upsample_factor) / upsample_factor
        region = np.ceil(upsample_factor * 1.5)
        dftshift = np.fix(region / 2.0)
        up = np.float64(upsample_factor)
        shifts = shifts + (res['fine'].astype(np.float64) - dftshift) / up
    ccmax = res['ccmax_re'] + 1j * res['ccmax_im']
    with np.errstate(all='ignore'):
        err = np.sqrt(np.abs(1.0 - ccmax * ccmax.conj() / (res['src_amp'] * res['tgt_amp'])))
    phase = np.arctan2(res['ccmax_im'], res['ccmax_re'])
    return shifts, err, phase


def register_pairs(tiles, pairs: np.ndarray, n0: int, n1: int, upsample_factor: int = 10,
                   normalization='phase', minmax=None):
    """Batched phase cross-correlation on a device tile stack [N, H, W].
    Returns (shifts [n,2], error [n], phasediff [n]) as numpy arrays."""
    code = _norm_code(normalization)
    if minmax is None:
        minmax = native.tile_minmax(tiles)
    res = native.register_pairs(tiles, minmax, pairs, n0, n1, upsample_factor, code)
    return shifts_from_results(res, upsample_factor)


def phase_cross_correlation(reference_image, moving_image, *, upsample_factor=1, normalization='phase',
                            device=None):
    """Drop-in for the call shape the reference uses (two equal-shape 2-D images already
    cropped): returns (shifts, error, phasediff).  Images go to the device as their own
    "tiles"; no min-max stretch is applied (identity range), like skimage."""
    import torch
    a = np.ascontiguousarray(reference_image)
    b = np.ascontiguousarray(moving_image)
    if a.shape != b.shape:
        raise ValueError("images must be same shape")
    if a.ndim != 2 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("device phase_cross_correlation takes 2-D uint8/uint16 images")
    dev = device or torch.device('cuda:0')
    tiles = torch.from_numpy(np.stack([a, b])).to(dev)
    # a (min > max) entry tells the kernel to use the pixels as they are
    minmax = torch.tensor([[1, 0], [1, 0]], dtype=torch.int32, device=dev)
    pairs = np.zeros(1, dtype=native.PAIR_DTYPE)
    pairs[0] = (0, 1, 0, 0, 0, 0)
    s, e, p = register_pairs(tiles, pairs, a.shape[0], a.shape[1], upsample_factor, normalization, minmax)
    return s[0], float(e[0]), float(p[0])


def horizontal_pair(ref_tile: int, mov_tile: int, height: int, width: int, max_overlap: int):
    n0, n1, (ry, rx), (my, mx) = placement.horizontal_crop_origins(height, width, max_overlap)
    return (ref_tile, mov_tile, ry, rx, my, mx), n0, n1


def vertical_pair(ref_tile: int, mov_tile: int, height: int, width: int, max_overlap: int):
    n0, n1, (ry, rx), (my, mx) = placement.vertical_crop_origins(height, width, max_overlap)
    return (ref_tile, mov_tile, ry, rx, my, mx), n0, n1


def horizontal_shift_from(shift: np.ndarray, n1: int) -> Tuple[int, int]:
    """(stitcher.py:511): python round() of numpy float64 -> banker's rounding."""
    return round(shift[0]), round(shift[1] - n1)


def vertical_shift_from(shift: np.ndarray, n0: int) -> Tuple[int, int]:
    """(stitcher.py:524)"""
    return round(shift[0] - n0), round(shift[1])


_SIDE_STREAMS = {}


def _side_stream(device):
    """One extra HIP stream per device for the second of two independent registration batches."""
    import torch
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class PendingShifts:
    """Centre-pair registration in flight on the device; ``result()`` synchronises and returns the Shifts."""

    def __init__(self, finish):
        self._finish, self._shifts = finish, None

    def result(self) -> Shifts:
        if self._finish is not None:
            self._shifts = self._finish()
            self._finish = None
        return self._shifts


def register_grid_center(tiles, n_rows: int, n_cols: int, xs: Sequence[float], ys: Sequence[float],
                         pixel_size_um: float, pixel_binning: int, normalization='phase',
                         scan_pattern: str = 'Unidirectional', tile_index=None) -> Shifts:
    """calculate_shifts (stitcher.py:422-498) on a device stack of one (channel, z) plane (synchronises)."""
    return register_grid_center_async(tiles, n_rows, n_cols, xs, ys, pixel_size_um, pixel_binning, normalization,
                                      scan_pattern, tile_index).result()


def register_grid_center_async(tiles, n_rows: int, n_cols: int, xs: Sequence[float], ys: Sequence[float],
                               pixel_size_um: float, pixel_binning: int, normalization='phase',
                               scan_pattern: str = 'Unidirectional', tile_index=None) -> PendingShifts:
    """Enqueue calculate_shifts (stitcher.py:422-498) on a device stack of one (channel, z) plane and return
    without synchronising: a caller with several independent regions enqueues the next region's
    registration ahead of the current region's fusion and collects it while that fusion runs.

    ``tile_index(row, col) -> index into tiles`` (default row-major).  Picks the centre tile
    and its right / bottom neighbours; S-Pattern adds the pair one row below.  Only the tiles
    involved are touched (min/max included), like the reference's get_tile calls."""
    height, width = int(tiles.shape[-2]), int(tiles.shape[-1])
    idx = tile_index or (lambda r, c: r * n_cols + c)
    mx, my = placement.registration_crop_widths(xs, ys, width, height, pixel_size_um, pixel_binning)
    ci, ri = (n_cols - 1) // 2, (n_rows - 1) // 2
    out = Shifts()
    hp, vp = [], []
    if ci + 1 < n_cols:
        hp.append((idx(ri, ci), idx(ri, ci + 1)))
        if scan_pattern == 'S-Pattern' and ri + 1 < n_rows:
            hp.append((idx(ri + 1, ci), idx(ri + 1, ci + 1)))
    if ri + 1 < n_rows:
        vp.append((idx(ri, ci), idx(ri + 1, ci)))
    used = sorted({t for p in hp + vp for t in p})
    if not used:
        return PendingShifts(lambda: out)
    local = {t: i for i, t in enumerate(used)}
    ptrs = native.pointer_table([tiles[t] for t in used], tiles.device)
    np_dtype = native.np_dtype_of_torch(tiles.dtype)
    minmax = native.tile_minmax(None, tile_ptrs=ptrs, shape=(height, width), np_dtype=np_dtype)
    code = _norm_code(normalization)

    def launch(pairs, make):
        rows = []
        for a, b in pairs:
            p, n0, n1 = make(local[a], local[b], height, width, mx if make is horizontal_pair else my)
            rows.append(p)
        pending = native.register_pairs_async(None, minmax, np.array(rows, dtype=native.PAIR_DTYPE), n0, n1, 10, code,
                                              tile_ptrs=ptrs, shape=(height, width), np_dtype=np_dtype)
        return pending, n0, n1

    # Both batches are enqueued before the first synchronising fetch, the vertical one on a side
    # stream: a batch of one or two pairs is a chain of small latency-bound kernels, and the two
    # chains are independent, so they run side by side.
    import torch
    hq = vq = None
    if hp and vp:
        main = torch.cuda.current_stream(tiles.device)
        side = _side_stream(tiles.device)
        ready = torch.cuda.Event()
        ready.record(main)                       # pointer table and min/max are on the main stream
        hq = launch(hp, horizontal_pair)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            vq = launch(vp, vertical_pair)
            done = torch.cuda.Event()
            done.record(side)
        main.wait_event(done)
    elif hp:
        hq = launch(hp, horizontal_pair)
    elif vp:
        vq = launch(vp, vertical_pair)
    def finish() -> Shifts:
        if hq:
            s = shifts_from_results(hq[0].fetch(), 10)[0]
            out.h_shift = horizontal_shift_from(s[0], hq[2])
            if len(hp) > 1:
                out.h_shift_rev = horizontal_shift_from(s[1], hq[2])
                out.h_shift_rev_odd = int(ri % 2 == 0)
        if scan_pattern == 'S-Pattern' and out.h_shift_rev is None:
            out.h_shift_rev = (0, 0)
        if vq:
            s = shifts_from_results(vq[0].fetch(), 10)[0]
            out.v_shift = vertical_shift_from(s[0], vq[1])
        return out

    return PendingShifts(finish)


def all_pairs(n_rows: int, n_cols: int, height: int, width: int, max_x_overlap: int, max_y_overlap: int,
              tile_index=None):
    """Every horizontally / vertically adjacent pair of a grid as two PAIR_DTYPE batches:
    (h_pairs, (n0, n1)), (v_pairs, (n0, n1))."""
    idx = tile_index or (lambda r, c: r * n_cols + c)
    hp, vp = [], []
    hshape = vshape = (0, 0)
    for r in range(n_rows):
        for c in range(n_cols - 1):
            p, n0, n1 = horizontal_pair(idx(r, c), idx(r, c + 1), height, width, max_x_overlap)
            hp.append(p)
            hshape = (n0, n1)
    for r in range(n_rows - 1):
        for c in range(n_cols):
            p, n0, n1 = vertical_pair(idx(r, c), idx(r + 1, c), height, width, max_y_overlap)
            vp.append(p)
            vshape = (n0, n1)
    return (np.array(hp, dtype=native.PAIR_DTYPE), hshape), (np.array(vp, dtype=native.PAIR_DTYPE), vshape)


# ---------------------------------------------------------------------------------------------------------------
# all-pairs registration sharded over the ranks (north star: "independent overlap pairs ... sharded across the 8 GPUs
# ... only the small global shift table all-gathered"; SURVEY.md 8e: n_pairs x {dy, dx, err} float64)
# ---------------------------------------------------------------------------------------------------------------
PAIR_H, PAIR_V = 0, 1


def grid_pair_list(n_rows: int, n_cols: int, present=None) -> List[Tuple[int, Tuple[int, int], Tuple[int, int]]]:
    """Every adjacent pair of an ``n_rows x n_cols`` grid as (kind, (r, c) reference cell, (r, c) moving cell), in
    TILE-ROW order: the horizontal pairs of row r, then the vertical pairs from row r to row r + 1, then row r + 1 ...
    (the reference's own pairs, stitcher.py:455-496, are the centre cell's two).  ``present``: a container of the
    cells that have a tile; pairs with a missing side are left out.  The order is what makes a contiguous run of
    the list touch a contiguous band of tile rows (``pairs_of_rank``)."""
    has = (lambda rc: True) if present is None else (lambda rc: rc in present)
    out = []
    for r in range(n_rows):
        for c in range(n_cols - 1):
            if has((r, c)) and has((r, c + 1)):
                out.append((PAIR_H, (r, c), (r, c + 1)))
        if r + 1 < n_rows:
            for c in range(n_cols):
                if has((r, c)) and has((r + 1, c)):
                    out.append((PAIR_V, (r, c), (r + 1, c)))
    return out


def pairs_of_rank(n_pairs: int, rank: int, world: int) -> List[int]:
    """Indices into ``grid_pair_list`` owned by ``rank``: ONE contiguous run.  With the list in tile-row order a rank
    then reads the tiles of its band of rows plus the row below it -- 1/world of the plane + one row -- where a
    block-cyclic deal of 2 x 1984 pair sides would make every rank of eight read about half of the plane."""
    from .sharding import contiguous_blocks
    return contiguous_blocks(n_pairs, rank, world)


def cells_of_pairs(pairs, indices) -> List[Tuple[int, int]]:
    """The grid cells (sorted) the pairs ``indices`` touch: what a rank has to read / upload / min-max."""
    return sorted({cell for i in indices for cell in pairs[i][1:]})


def register_pair_subset(tiles, local_index, pairs, indices, height: int, width: int, max_x_overlap: int,
                         max_y_overlap: int, normalization='phase', minmax=None) -> np.ndarray:
    """Register the pairs ``indices`` of ``pairs`` on a device stack ``tiles`` that holds (at least) the cells those
    pairs touch; ``local_index[(r, c)]`` is a cell's position in the stack.  One batch per direction (their crop
    shapes differ).  Returns [len(indices), 3] float64 rows {dy, dx, err}: skimage's raw sub-pixel shift of the crop
    pair (before the reference's ``round`` and ``- w``, stitcher.py:511,524) and its error."""
    out = np.full((len(indices), 3), np.nan, dtype=np.float64)
    if not len(indices):
        return out
    if minmax is None:
        minmax = native.tile_minmax(tiles)      # enqueued; the pair records below are made while it runs
    code = _norm_code(normalization)
    kinds = np.fromiter((pairs[i][0] for i in indices), dtype=np.int64, count=len(indices))
    ref = np.fromiter((local_index[pairs[i][1]] for i in indices), dtype=np.int64, count=len(indices))
    mov = np.fromiter((local_index[pairs[i][2]] for i in indices), dtype=np.int64, count=len(indices))
    pending = []
    # both directions are enqueued before either is read back (one host round trip for the two batches)
    for kind, origins, ov in ((PAIR_H, placement.horizontal_crop_origins, max_x_overlap),
                              (PAIR_V, placement.vertical_crop_origins, max_y_overlap)):
        slots = np.flatnonzero(kinds == kind)
        if not len(slots):
            continue
        n0, n1, (ry, rx), (my, mx) = origins(height, width, int(ov))
        recs = np.zeros(len(slots), dtype=native.PAIR_DTYPE)
        for name, values in zip(native.PAIR_DTYPE.names, (ref[slots], mov[slots], ry, rx, my, mx)):
            recs[name] = values
        pending.append((slots, native.register_pairs_async(tiles, minmax, recs, n0, n1, 10, code)))
    for slots, job in pending:
        shifts, err, _ = shifts_from_results(job.fetch(), 10)
        out[slots, 0:2] = shifts
        out[slots, 2] = err
    return out


def pair_table_medians(pairs, table: np.ndarray, height: int, width: int, max_x_overlap: int, max_y_overlap: int,
                       n_rows: int, scan_pattern: str = 'Unidirectional') -> dict:
    """The reference's state (h_shift, v_shift[, h_shift_rev, h_shift_rev_odd]) from the pair table: every pair's
    shift converted exactly as the reference converts its centre pair's (python ``round``, ``- w``;
    stitcher.py:511,524), then the per-axis LOWER median (stays an integer) over the pairs of a direction; with an
    S-Pattern the rows of the centre row's parity give h_shift and the others h_shift_rev (:486-496).  Pure host
    arithmetic on the gathered table, so every rank arrives at the same integers.  Returns only the keys a pair
    was found for (a grid of one row has no v_shift), like the reference leaves the others at their defaults."""
    def med2(rows):      # per-axis LOWER median of [k, 2] integers
        srt = np.sort(rows, axis=0)
        return (int(srt[(len(rows) - 1) // 2, 0]), int(srt[(len(rows) - 1) // 2, 1]))

    table = np.asarray(table, dtype=np.float64).reshape(len(pairs), -1)
    kinds = np.fromiter((p[0] for p in pairs), dtype=np.int64, count=len(pairs))
    ref_row = np.fromiter((p[1][0] for p in pairs), dtype=np.int64, count=len(pairs))
    h_n1 = placement.horizontal_crop_origins(height, width, int(max_x_overlap))[1] if (kinds == PAIR_H).any() else 0
    v_n0 = placement.vertical_crop_origins(height, width, int(max_y_overlap))[0] if (kinds == PAIR_V).any() else 0
    cy = (n_rows - 1) // 2
    s_pattern = scan_pattern == 'S-Pattern'
    found = np.isfinite(table[:, 0]) & np.isfinite(table[:, 1])
    # python round() of a float64 and numpy.rint round halves to even alike (horizontal_shift_from / vertical_shift_from,
    # stitcher.py:511,524, for the whole table at once)
    clean = np.where(found[:, None], table[:, :2], 0.0)
    h_all = np.stack([np.rint(clean[:, 0]), np.rint(clean[:, 1] - h_n1)], axis=1).astype(np.int64)
    v_all = np.stack([np.rint(clean[:, 0] - v_n0), np.rint(clean[:, 1])], axis=1).astype(np.int64)
    is_rev = s_pattern & (ref_row % 2 != cy % 2)
    fwd = h_all[found & (kinds == PAIR_H) & ~is_rev]
    rev = h_all[found & (kinds == PAIR_H) & is_rev]
    ver = v_all[found & (kinds == PAIR_V)]
    out = {}
    if len(fwd):
        out['h_shift'] = med2(fwd)
    if len(rev):
        out['h_shift_rev'] = med2(rev)
        out['h_shift_rev_odd'] = int(cy % 2 == 0)
    if len(ver):
        out['v_shift'] = med2(ver)
    return out


def shifts_from_pair_table(pairs, table, height, width, max_x_overlap, max_y_overlap, n_rows,
                           scan_pattern: str = 'Unidirectional') -> Shifts:
    """``pair_table_medians`` as a Shifts (absent directions at the reference's defaults)."""
    return Shifts(**pair_table_medians(pairs, table, height, width, max_x_overlap, max_y_overlap, n_rows, scan_pattern))


def register_all_pairs_sharded(pairs, load_cells, height: int, width: int, max_x_overlap: int, max_y_overlap: int,
                               normalization='phase', rank: int = 0, world: int = 1, device=None, group=None) -> np.ndarray:
    """All-pairs registration with the pairs dealt over the ranks.  ``pairs`` = ``grid_pair_list(...)`` (identical on
    every rank); ``load_cells(cells) -> device stack [len(cells), H, W]`` brings in exactly the tiles THIS rank's
    pairs touch (files -> H2D in the product, the device generator in the bench).  Every rank registers its run of
    pairs; the [n_pairs, 3] float64 table {dy, dx, err} is all-gathered (RCCL over xGMI with the nccl backend; the
    only collective) and returned whole on every rank.  Nothing else is exchanged."""
    from . import sharding
    mine = pairs_of_rank(len(pairs), rank, world)
    cells = cells_of_pairs(pairs, mine)
    local = np.zeros((0, 3), dtype=np.float64)
    if mine:
        tiles = load_cells(cells)
        local = register_pair_subset(tiles, {c: i for i, c in enumerate(cells)}, pairs, mine, height, width,
                                     max_x_overlap, max_y_overlap, normalization)
    return sharding.all_gather_pair_table(local, len(pairs), rank, world, device=device, group=group)


def consensus_shift(shifts: np.ndarray, errors: np.ndarray) -> Tuple[float, float]:
    """Robust per-axis median of a batch of pair shifts (all-pairs extension)."""
    ok = np.isfinite(errors)
    s = shifts[ok] if ok.any() else shifts
    return float(np.median(s[:, 0])), float(np.median(s[:, 1]))

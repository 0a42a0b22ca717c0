"""Integer geometry of a stitched region, computed on the host exactly as the reference does.

Everything here is Python ``int``/``float`` arithmetic with the reference's own operators
(floor division on negatives, ``int()`` truncation, banker's ``round``), because integer
placements are part of the parity contract.  The device only ever sees
the resulting rectangles (``sq_rect``) and crop origins (``sq_pair``).

Reference: stitcher.py:298-354 (canvas size), :444-452 (registration crop widths),
:656-679 (placement), :570-587 (half-overlap crop).
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

Shift = Tuple[int, int]   # (dy, dx)


@dataclasses.dataclass
class Shifts:
    """Registration result applied to every tile of every region (stitcher.py:113-117)."""
    h_shift: Shift = (0, 0)
    v_shift: Shift = (0, 0)
    h_shift_rev: Optional[Shift] = None    # S-Pattern only
    h_shift_rev_odd: int = 0               # rows with row % 2 == this use h_shift_rev

    def horizontal_for_row(self, row: int) -> Shift:
        """(stitcher.py:571-574, :660-663)"""
        if self.h_shift_rev is not None and row % 2 == self.h_shift_rev_odd:
            return self.h_shift_rev
        return self.h_shift


def registration_crop_widths(xs: Sequence[float], ys: Sequence[float], width: int, height: int,
                             pixel_size_um: float, pixel_binning: int) -> Tuple[int, int]:
    """(max_x_overlap, max_y_overlap): ``round(|W - dx_px| * 1.05) // 2 * binning``
    (stitcher.py:444-452)."""
    dx_px = (xs[1] - xs[0]) * 1000 / pixel_size_um
    dy_px = (ys[1] - ys[0]) * 1000 / pixel_size_um
    return (round(abs(width - dx_px) * 1.05) // 2 * pixel_binning,
            round(abs(height - dy_px) * 1.05) // 2 * pixel_binning)


def horizontal_crop_origins(height: int, width: int, max_overlap: int):
    """Crop of calculate_horizontal_shift (stitcher.py:504-506) as
    (n0, n1, (ref_y0, ref_x0), (mov_y0, mov_x0)) with python slice semantics."""
    margin = int(height * 0.25)
    rows = range(height)[margin:-margin]
    ref_cols = range(width)[-max_overlap:]
    mov_cols = range(width)[:max_overlap]
    if len(rows) == 0 or len(ref_cols) != len(mov_cols) or len(ref_cols) == 0:
        raise ValueError("images must be same shape")     # what skimage raises on mismatched crops
    return len(rows), len(ref_cols), (rows[0], ref_cols[0]), (rows[0], mov_cols[0])


def vertical_crop_origins(height: int, width: int, max_overlap: int):
    """Crop of calculate_vertical_shift (stitcher.py:517-519)."""
    margin = int(width * 0.25)
    cols = range(width)[margin:-margin]
    ref_rows = range(height)[-max_overlap:]
    mov_rows = range(height)[:max_overlap]
    if len(cols) == 0 or len(ref_rows) != len(mov_rows) or len(ref_rows) == 0:
        raise ValueError("images must be same shape")
    return len(ref_rows), len(cols), (ref_rows[0], cols[0]), (mov_rows[0], cols[0])


def canvas_size(n_cols: int, n_rows: int, width: int, height: int, *, use_registration: bool,
                shifts: Optional[Shifts] = None, xs: Sequence[float] = (), ys: Sequence[float] = (),
                pixel_size_um: float = 1.0) -> Tuple[int, int]:
    """(width_pixels, height_pixels) of the region canvas (stitcher.py:318-343).

    Registered mode reproduces the reference's formula verbatim, including the signed
    ``height - v_shift[0]`` (a negative v_shift[0] ADDS the overlap, so the canvas is taller
    than the tiles need; the extra rows stay zero)."""
    if use_registration:
        s = shifts or Shifts()
        if s.h_shift_rev is not None:
            max_h = (max(abs(s.h_shift[0]), abs(s.h_shift_rev[0])), max(abs(s.h_shift[1]), abs(s.h_shift_rev[1])))
        else:
            max_h = (abs(s.h_shift[0]), abs(s.h_shift[1]))
        w_px = int(width + ((n_cols - 1) * (width - max_h[1])))
        w_px += abs((n_rows - 1) * s.v_shift[1])
        h_px = int(height + ((n_rows - 1) * (height - s.v_shift[0])))
        h_px += abs((n_cols - 1) * max_h[0])
        return w_px, h_px
    w_mm = max(xs) - min(xs) + (width * pixel_size_um / 1000)
    h_mm = max(ys) - min(ys) + (height * pixel_size_um / 1000)
    return int(np.ceil(w_mm * 1000 / pixel_size_um)), int(np.ceil(h_mm * 1000 / pixel_size_um))


def pyramid_levels(width_pixels: int, height_pixels: int, max_dimension: int = 1) -> int:
    """(stitcher.py:346-352)"""
    return max(1, math.ceil(np.log2(max(width_pixels, height_pixels) / 1024 * max_dimension)))


def registered_rect(row: int, col: int, n_rows: int, n_cols: int, width: int, height: int,
                    shifts: Shifts, crop: bool = True) -> Tuple[int, int, int, int, int, int]:
    """sq_rect fields (src_y0, src_x0, h, w, dst_y, dst_x) of the tile at grid (row, col)
    (stitcher.py:665-676 then :577-587).  ``crop=False`` keeps the full tile (feather mode)."""
    h_shift = shifts.horizontal_for_row(row)
    v_shift = shifts.v_shift
    x_px = int(col * (width + h_shift[1]))
    y_px = int(row * (height + v_shift[0]))
    if h_shift[0] < 0:
        y_px += int((n_cols - 1 - col) * abs(h_shift[0]))
    else:
        y_px += int(col * h_shift[0])
    if v_shift[1] < 0:
        x_px += int((n_rows - 1 - row) * abs(v_shift[1]))
    else:
        x_px += int(row * v_shift[1])
    top = bottom = left = right = 0
    if crop:
        v_crop = max(0, (-v_shift[0] // 2) - abs(h_shift[0]) // 2)
        h_crop = max(0, (-h_shift[1] // 2) - abs(v_shift[1]) // 2)
        top = v_crop if row > 0 else 0
        bottom = v_crop if row < n_rows - 1 else 0
        left = h_crop if col > 0 else 0
        right = h_crop if col < n_cols - 1 else 0
    # python slicing tile[top:H-bottom, left:W-right] of the reference (empty when crops cross)
    h = max(0, height - bottom - top)
    w = max(0, width - right - left)
    return top, left, h, w, y_px + top, x_px + left


def coordinate_rect(x_mm: float, y_mm: float, x_min: float, y_min: float, width: int, height: int,
                    pixel_size_um: float) -> Tuple[int, int, int, int, int, int]:
    """Coordinate-only placement (stitcher.py:678-679): truncating mm -> px, no crop."""
    return (0, 0, height, width,
            int((y_mm - y_min) * 1000 / pixel_size_um), int((x_mm - x_min) * 1000 / pixel_size_um))


def grid_rects(n_rows: int, n_cols: int, width: int, height: int, shifts: Shifts,
               order: Optional[Sequence[Tuple[int, int]]] = None, crop: bool = True) -> np.ndarray:
    """[n, 6] rectangles of a full registered grid in write ``order`` (list of (row, col);
    default row-major).  Same integers as ``registered_rect`` per tile, computed for all tiles at
    once in int64 (every product and sum of the reference's formula is an exact integer)."""
    if order is None:
        rows, cols = np.divmod(np.arange(n_rows * n_cols, dtype=np.int64), n_cols)
    else:
        rc = np.asarray(order, dtype=np.int64).reshape(-1, 2)
        rows, cols = rc[:, 0], rc[:, 1]
    v_shift = shifts.v_shift
    # per-row horizontal shift (S-Pattern rows use h_shift_rev)
    hs = np.array([shifts.horizontal_for_row(int(r)) for r in range(n_rows)], dtype=np.int64).reshape(-1, 2)
    h0, h1 = hs[rows, 0], hs[rows, 1]
    x_px = cols * (width + h1)
    y_px = rows * (height + v_shift[0])
    y_px = y_px + np.where(h0 < 0, (n_cols - 1 - cols) * np.abs(h0), cols * h0)
    x_px = x_px + ((n_rows - 1 - rows) * abs(v_shift[1]) if v_shift[1] < 0 else rows * v_shift[1])
    zeros = np.zeros_like(rows)
    top = bottom = left = right = zeros
    if crop:
        # python's floor division on negatives is numpy's too
        v_crop = np.maximum(0, (-v_shift[0] // 2) - np.abs(h0) // 2)
        h_crop = np.maximum(0, (-h1 // 2) - abs(v_shift[1]) // 2)
        top = np.where(rows > 0, v_crop, 0)
        bottom = np.where(rows < n_rows - 1, v_crop, 0)
        left = np.where(cols > 0, h_crop, 0)
        right = np.where(cols < n_cols - 1, h_crop, 0)
    h = np.maximum(0, height - bottom - top)
    w = np.maximum(0, width - right - left)
    return np.stack([top, left, h, w, y_px + top, x_px + left], axis=1).astype(np.int64)


def filename_order(fovs: Sequence[int]) -> List[int]:
    """Positions of ``fovs`` in the order the reference meets them inside one (z, channel)
    plane: file names ``{region}_{fov}_{z}_{channel}`` sort as strings (stitcher.py:168), and
    '_' sorts after the digits, so fov 10 comes before fov 1, which comes before fov 2."""
    return sorted(range(len(fovs)), key=lambda i: f"{fovs[i]}_")


def clip_rect_to_rows(rect, y0: int, y1: int):
    """A tile rectangle (src_y0, src_x0, h, w, dst_y, dst_x) cut to the canvas rows [y0, y1) and re-based to row y0
    (the canvas of one row band, sharding.row_bands) -- or None when the tile does not reach into the band."""
    sy, sx, h, w, dy, dx = (int(v) for v in rect)
    top, bottom = max(dy, int(y0)), min(dy + h, int(y1))
    if bottom <= top or w <= 0:
        return None
    return (sy + (top - dy), sx, bottom - top, w, top - int(y0), dx)


def check_rect_fits_like_numpy(rect, canvas_h: int, canvas_w: int) -> None:
    """The reference clips a tile to the canvas with ``tile[:y_end - y, :x_end - x]`` (stitcher.py:589-598).  When a
    tile starts beyond the canvas edge the slice end is negative, python counts it from the tile's other end,
    and the assignment dies with numpy's broadcast ValueError.  Garbage shifts only -- but the same input must
    fail the same way here, so this re-enacts the two slices and raises that error."""
    _, _, h, w, dy, dx = (int(v) for v in rect)
    y_end, x_end = min(dy + h, canvas_h), min(dx + w, canvas_w)
    tile_rows, tile_cols = len(range(h)[:y_end - dy]), len(range(w)[:x_end - dx])
    dst_rows, dst_cols = len(range(canvas_h)[dy:y_end]), len(range(canvas_w)[dx:x_end])
    if (tile_rows, tile_cols) != (dst_rows, dst_cols):
        raise ValueError(f"could not broadcast input array from shape ({tile_rows},{tile_cols}) into shape ({dst_rows},{dst_cols})")

"""CPU oracle for the registration-and-fusion hot path -- TEST INFRASTRUCTURE ONLY.

A numpy restatement of what the reference computes on the path SURVEY.md section 8
scopes.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker; the product
(``image-stitcher_amd/``) never imports it and has no CPU fallback.

Pinning: every function below is checked against vectors produced by executing the
unmodified reference (``tests/golden/make_golden.py``, run under the authoring
container's python3.9 + scikit-image 0.18.3) -- see ``tests/test_oracle_golden.py``.
``normalization='phase'`` (the scikit-image >= 0.19 default, which cannot be
installed offline) is pinned by driving the real 0.18.3 peak search / upsampled
refinement with the phase-normalised cross-power spectrum.
Feather fusion is an extension the reference does not have: parity unpinned by the
reference for that mode; this file is its definition.

Cited lines are ``/root/reference/stitcher.py`` unless they say ``skimage``, which is
``skimage/registration/_phase_cross_correlation.py`` of scikit-image 0.18.3.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# registration
# --------------------------------------------------------------------------------------


def normalize_image(img: np.ndarray, dtype) -> np.ndarray:
    """Min-max stretch to the full range of ``dtype`` with a truncating cast
    (stitcher.py:613-617).  max == min gives 0/0 -> NaN -> cast, as in the reference."""
    img = np.asarray(img)
    lo, hi = img.min(), img.max()
    with np.errstate(all='ignore'):
        unit = (img - lo) / (hi - lo)
        scale = np.iinfo(dtype).max if np.issubdtype(dtype, np.integer) else 1
        return (unit * scale).astype(dtype)


def _upsampled_dft(data: np.ndarray, region: int, upsample: float, offsets: Sequence[float]) -> np.ndarray:
    """Matrix-multiply DFT of a ``region``-wide neighbourhood (skimage :11-75):
    last axis first; each pass contracts the array's last axis and puts the new
    axis in front, so a 2-D input comes back as [axis0, axis1]."""
    for n, off in list(zip(data.shape, offsets))[::-1]:
        phase = (np.arange(region) - off)[:, None] * np.fft.fftfreq(n, upsample)
        data = np.tensordot(np.exp(-2j * np.pi * phase), data, axes=(1, -1))
    return data


def phase_cross_correlation(ref: np.ndarray, mov: np.ndarray, upsample_factor: int = 10,
                            normalization: Optional[str] = 'phase', space: str = 'real'):
    """Sub-pixel translation by upsampled cross-correlation (skimage :109-276).

    Returns ``(shifts[2] float64, error, phasediff, detail)`` where ``detail`` carries the
    integer peak and the refinement index the device path reports.
    ``normalization``: ``None`` = scikit-image 0.18 behaviour; ``'phase'`` = the one-line
    addition of >= 0.19 (divide the cross-power spectrum by max(|.|, 100 eps)).
    """
    if ref.shape != mov.shape:
        raise ValueError("images must be same shape")
    if space == 'real':
        f = np.fft.fftn(ref)                                   # skimage :204
        g = np.fft.fftn(mov)                                   # skimage :205
    else:
        f, g = ref, mov
    shape = f.shape
    prod = f * g.conj()                                        # skimage :211
    if normalization == 'phase':
        eps = np.finfo(prod.real.dtype).eps
        prod = prod / np.maximum(np.abs(prod), 100 * eps)
    elif normalization is not None:
        raise ValueError("normalization must be either phase or None")
    cc = np.fft.ifftn(prod)                                    # skimage :212
    peak = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)   # skimage :215
    mid = np.array([np.fix(n / 2) for n in shape])
    shifts = np.stack(peak).astype(np.float64)
    wrap = shifts > mid
    shifts[wrap] -= np.array(shape)[wrap]                      # skimage :217-220
    coarse = shifts.copy()
    fine = None
    if upsample_factor == 1:
        ccmax = cc[peak]
        src_amp = np.sum(np.real(f * f.conj())) / f.size
        tgt_amp = np.sum(np.real(g * g.conj())) / g.size
    else:
        shifts = np.round(shifts * upsample_factor) / upsample_factor     # skimage :232
        region = int(np.ceil(upsample_factor * 1.5))                      # skimage :233
        dftshift = np.fix(region / 2.0)                                   # skimage :235
        up = np.float64(upsample_factor)
        offs = dftshift - shifts * up                                     # skimage :238
        cc_up = _upsampled_dft(prod.conj(), region, up, offs).conj()      # skimage :239-242
        fine = np.unravel_index(np.argmax(np.abs(cc_up)), cc_up.shape)    # skimage :244
        ccmax = cc_up[fine]
        shifts = shifts + (np.stack(fine).astype(np.float64) - dftshift) / up   # skimage :248-250
        src_amp = np.sum(np.real(f * f.conj()))
        tgt_amp = np.sum(np.real(g * g.conj()))
    for d in range(f.ndim):                                               # skimage :256-258
        if shape[d] == 1:
            shifts[d] = 0
    with np.errstate(all='ignore'):
        err = np.sqrt(np.abs(1.0 - ccmax * ccmax.conj() / (src_amp * tgt_amp)))   # skimage :91-106
    phasediff = math.atan2(ccmax.imag, ccmax.real)
    detail = dict(coarse=[int(v) for v in coarse], fine=None if fine is None else [int(v) for v in fine],
                  ccmax_abs=float(abs(ccmax)), src_amp=float(src_amp), tgt_amp=float(tgt_amp))
    return shifts, float(err), float(phasediff), detail


def overlap_crops_horizontal(left: np.ndarray, right: np.ndarray, max_overlap: int):
    """Centre-half rows, last/first ``max_overlap`` columns (stitcher.py:504-506)."""
    margin = int(left.shape[0] * 0.25)
    return left[margin:-margin, -max_overlap:], right[margin:-margin, :max_overlap]


def overlap_crops_vertical(top: np.ndarray, bot: np.ndarray, max_overlap: int):
    """Last/first ``max_overlap`` rows, centre-half columns (stitcher.py:517-519)."""
    margin = int(top.shape[1] * 0.25)
    return top[-max_overlap:, margin:-margin], bot[:max_overlap, margin:-margin]


def calculate_horizontal_shift(left, right, max_overlap, dtype, normalization='phase'):
    """(stitcher.py:500-511): normalise both tiles, crop, register, python round()."""
    a, b = overlap_crops_horizontal(normalize_image(left, dtype), normalize_image(right, dtype), max_overlap)
    s = phase_cross_correlation(a, b, upsample_factor=10, normalization=normalization)[0]
    return round(s[0]), round(s[1] - a.shape[1])


def calculate_vertical_shift(top, bot, max_overlap, dtype, normalization='phase'):
    """(stitcher.py:513-524)."""
    a, b = overlap_crops_vertical(normalize_image(top, dtype), normalize_image(bot, dtype), max_overlap)
    s = phase_cross_correlation(a, b, upsample_factor=10, normalization=normalization)[0]
    return round(s[0] - a.shape[0]), round(s[1])


# --------------------------------------------------------------------------------------
# acquisition metadata (just enough to drive the path from a folder)
# --------------------------------------------------------------------------------------

_EXT = ('.bmp', '.tiff', 'tif', 'jpg', 'jpeg', 'png')   # NB 'tif' etc. without a dot, as the reference


class Acquisition:
    """What parse_acquisition_metadata leaves behind (stitcher.py:121-257): an ordered
    mapping (t, region, fov, z, channel) -> tile record, in sorted-filename order."""

    def __init__(self):
        self.meta: Dict[tuple, dict] = {}
        self.timepoints: List[str] = []
        self.regions: List[str] = []
        self.channel_names: List[str] = []
        self.monochrome_channels: List[str] = []
        self.num_z = 1
        self.pixel_size_um = 0.0
        self.pixel_binning = 1
        self.input_height = 0
        self.input_width = 0
        self.dtype = np.uint16

    def region_data(self, t, region) -> Dict[tuple, dict]:
        """(stitcher.py:260-280)"""
        t = int(t)
        data = {k: v for k, v in self.meta.items() if k[0] == t and k[1] == region}
        if not data:
            raise ValueError(f"No data found for timepoint {t}, region {region}")
        return data


def parse_acquisition(folder: str, read_image) -> Acquisition:
    """Folder -> Acquisition (stitcher.py:121-257).  ``read_image(path) -> ndarray``."""
    acq = Acquisition()
    acq.timepoints = sorted([d for d in os.listdir(folder)
                             if os.path.isdir(os.path.join(folder, d)) and d.isdigit()], key=int)
    with open(os.path.join(folder, 'acquisition parameters.json')) as fh:
        ap = json.load(fh)
    focal = ap['objective']['tube_lens_f_mm'] / ap['objective']['magnification']
    acq.pixel_size_um = ap['sensor_pixel_size_um'] / (ap['tube_lens_mm'] / focal)   # :131-139
    acq.pixel_binning = ap.get('pixel_binning', 1)
    regions, channels, max_z = set(), set(), 0
    for tp in acq.timepoints:
        tdir = os.path.join(folder, tp)
        try:
            with open(os.path.join(tdir, 'coordinates.csv')) as fh:
                header = fh.readline().strip().split(',')
                rows = [dict(zip(header, ln.strip().split(','))) for ln in fh if ln.strip()]
        except FileNotFoundError:
            continue
        files = sorted(f for f in os.listdir(tdir) if f.endswith(_EXT) and 'focus_camera' not in f)  # :168
        for f in files:
            region, fov, z, rest = f.split('_', 3)                              # :172
            fov, z = int(fov), int(z)
            channel = os.path.splitext(rest)[0].replace('_', ' ').replace('full ', 'full_')   # :174
            row = next((r for r in rows if r['region'] == region and int(r['fov']) == fov
                        and int(r['z_level']) == z), None)                      # :176-186 (first match)
            if row is None:
                continue
            acq.meta[(int(tp), region, fov, z, channel)] = dict(
                filepath=os.path.join(tdir, f), x=float(row['x (mm)']), y=float(row['y (mm)']),
                z=float(row['z (um)']), channel=channel, z_level=z, region=region, fov_idx=fov, t=int(tp))
            regions.add(region)
            channels.add(channel)
            max_z = max(max_z, z)
    acq.regions = sorted(regions)
    acq.channel_names = sorted(channels)
    acq.num_z = max_z + 1
    first_rec = next(iter(acq.meta.values()))
    first = read_image(first_rec['filepath'])
    acq.dtype = first.dtype.type
    acq.input_height, acq.input_width = first.shape[:2]
    # RGB files become three monochrome channels (stitcher.py:238-246); like the reference this
    # needs the first (t, region, fov, z) to exist for every channel (KeyError otherwise)
    for channel in acq.channel_names:
        key = (first_rec['t'], first_rec['region'], first_rec['fov_idx'], first_rec['z_level'], channel)
        img = read_image(acq.meta[key]['filepath'])
        if img.ndim == 3 and img.shape[2] == 3:
            base = channel.split('_')[0]
            acq.monochrome_channels.extend([f"{base}_R", f"{base}_G", f"{base}_B"])
        else:
            acq.monochrome_channels.append(channel)
    return acq


# --------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------


def grid_positions(region_data: Dict[tuple, dict]) -> Tuple[List[float], List[float]]:
    """Sorted unique stage x / y (stitcher.py:315-316)."""
    xs = sorted(set(v['x'] for v in region_data.values()))
    ys = sorted(set(v['y'] for v in region_data.values()))
    return xs, ys


def max_overlaps(xs, ys, width, height, pixel_size_um, pixel_binning) -> Tuple[int, int]:
    """Registration crop widths (stitcher.py:444-452) -> (max_x_overlap, max_y_overlap)."""
    dx_px = (xs[1] - xs[0]) * 1000 / pixel_size_um
    dy_px = (ys[1] - ys[0]) * 1000 / pixel_size_um
    mx = round(abs(width - dx_px) * 1.05) // 2 * pixel_binning
    my = round(abs(height - dy_px) * 1.05) // 2 * pixel_binning
    return mx, my


def output_dimensions(xs, ys, width, height, pixel_size_um, use_registration, h_shift=(0, 0),
                      v_shift=(0, 0), h_shift_rev=None, n_regions_grid: int = 1):
    """(Wc, Hc, num_pyramid_levels) (stitcher.py:318-354), quirks included: the height uses
    ``height - v_shift[0]`` with the signed v_shift[0]."""
    if use_registration:
        ncols, nrows = len(xs), len(ys)
        if h_shift_rev is not None:
            mh = (max(abs(h_shift[0]), abs(h_shift_rev[0])), max(abs(h_shift[1]), abs(h_shift_rev[1])))
        else:
            mh = (abs(h_shift[0]), abs(h_shift[1]))
        wc = int(width + ((ncols - 1) * (width - mh[1])))
        wc += abs((nrows - 1) * v_shift[1])
        hc = int(height + ((nrows - 1) * (height - v_shift[0])))
        hc += abs((ncols - 1) * mh[0])
    else:
        w_mm = max(xs) - min(xs) + (width * pixel_size_um / 1000)
        h_mm = max(ys) - min(ys) + (height * pixel_size_um / 1000)
        wc = int(np.ceil(w_mm * 1000 / pixel_size_um))
        hc = int(np.ceil(h_mm * 1000 / pixel_size_um))
    levels = max(1, math.ceil(np.log2(max(wc, hc) / 1024 * n_regions_grid)))
    return wc, hc, levels


def tile_rect(tile_info: dict, xs, ys, width, height, pixel_size_um, use_registration,
              h_shift, v_shift, h_shift_rev, h_shift_rev_odd, canvas_w, canvas_h):
    """Where one tile lands: (src_y0, src_x0, h, w, dst_y, dst_x) or None if nothing is
    written (stitcher.py:656-679 placement, :570-594 crop and clip)."""
    if use_registration:
        col = xs.index(tile_info['x'])
        row = ys.index(tile_info['y'])
        h = h_shift
        if h_shift_rev is not None and row % 2 == h_shift_rev_odd:
            h = h_shift_rev
        x_px = int(col * (width + h[1]))
        y_px = int(row * (height + v_shift[0]))
        if h[0] < 0:
            y_px += int((len(xs) - 1 - col) * abs(h[0]))
        else:
            y_px += int(col * h[0])
        if v_shift[1] < 0:
            x_px += int((len(ys) - 1 - row) * abs(v_shift[1]))
        else:
            x_px += int(row * v_shift[1])
        vcrop = max(0, (-v_shift[0] // 2) - abs(h[0]) // 2)
        hcrop = max(0, (-h[1] // 2) - abs(v_shift[1]) // 2)
        top = vcrop if row > 0 else 0
        bottom = vcrop if row < len(ys) - 1 else 0
        left = hcrop if col > 0 else 0
        right = hcrop if col < len(xs) - 1 else 0
    else:
        x_px = int((tile_info['x'] - min(xs)) * 1000 / pixel_size_um)
        y_px = int((tile_info['y'] - min(ys)) * 1000 / pixel_size_um)
        top = bottom = left = right = 0
    return x_px, y_px, top, bottom, left, right


def apply_flatfield(tile: np.ndarray, flatfield: Optional[np.ndarray], dtype) -> np.ndarray:
    """divide -> clip -> truncating cast (stitcher.py:607-611); None = no flatfield for
    that channel = identity."""
    if flatfield is None:
        return tile
    info = np.iinfo(dtype)
    with np.errstate(all='ignore'):
        return (tile / flatfield).clip(min=info.min, max=info.max).astype(dtype)


def place_tile(canvas: np.ndarray, c: int, z: int, tile: np.ndarray, x_px, y_px, top, bottom, left, right):
    """Python-slice semantics of stitcher.py:583-598 on a numpy canvas (overwrite)."""
    tile = tile[top:tile.shape[0] - bottom, left:tile.shape[1] - right]
    x_px += left
    y_px += top
    y_end = min(y_px + tile.shape[0], canvas.shape[3])
    x_end = min(x_px + tile.shape[1], canvas.shape[4])
    canvas[0, c, z, y_px:y_end, x_px:x_end] = tile[:y_end - y_px, :x_end - x_px]


class RegionPlan:
    """Everything integer about one (t, region): canvas size and per-file rectangles,
    in the reference's write order."""

    def __init__(self):
        self.canvas_w = 0
        self.canvas_h = 0
        self.levels = 1
        self.files: List[dict] = []     # filepath, c, z, x_px, y_px, top, bottom, left, right


def plan_region(acq: Acquisition, t, region, use_registration, h_shift=(0, 0), v_shift=(0, 0),
                h_shift_rev=None, h_shift_rev_odd=0, n_regions_grid=1) -> RegionPlan:
    rd = acq.region_data(t, region)
    xs, ys = grid_positions(rd)
    plan = RegionPlan()
    plan.canvas_w, plan.canvas_h, plan.levels = output_dimensions(
        xs, ys, acq.input_width, acq.input_height, acq.pixel_size_um, use_registration,
        h_shift, v_shift, h_shift_rev, n_regions_grid)
    for key, info in rd.items():
        _, _, fov, z, channel = key
        x_px, y_px, top, bottom, left, right = tile_rect(
            info, xs, ys, acq.input_width, acq.input_height, acq.pixel_size_um, use_registration,
            h_shift, v_shift, h_shift_rev, h_shift_rev_odd, plan.canvas_w, plan.canvas_h)
        if channel in acq.monochrome_channels:
            targets = [(acq.monochrome_channels.index(channel), -1)]
        else:   # RGB file: one placement per colour plane (stitcher.py:551-556)
            base = channel.split('_')[0]
            targets = [(acq.monochrome_channels.index(f"{base}_{col}"), i) for i, col in enumerate('RGB')]
        for c_idx, rgb in targets:
            plan.files.append(dict(filepath=info['filepath'], c=c_idx, rgb=rgb, z=z, fov=fov,
                                   x_px=x_px, y_px=y_px, top=top, bottom=bottom, left=left, right=right))
    return plan


def calculate_shifts(acq: Acquisition, t, region, read_image, registration_channel='',
                     registration_z_level=0, scan_pattern='Unidirectional', normalization='phase'):
    """Centre-pair registration (stitcher.py:422-498) ->
    dict(h_shift, v_shift[, h_shift_rev, h_shift_rev_odd])."""
    rd = acq.region_data(t, region)
    xs, ys = grid_positions(rd)
    if not registration_channel or registration_channel not in acq.channel_names:
        registration_channel = acq.channel_names[0]
    mx, my = max_overlaps(xs, ys, acq.input_width, acq.input_height, acq.pixel_size_um, acq.pixel_binning)

    def get_tile(x, y):
        for v in rd.values():
            if v['x'] == x and v['y'] == y and v['channel'] == registration_channel \
                    and v['z_level'] == registration_z_level:
                return read_image(v['filepath'])
        return None

    ci, ri = (len(xs) - 1) // 2, (len(ys) - 1) // 2
    out = dict(h_shift=(0, 0), v_shift=(0, 0))
    right_x = bottom_y = None
    if ci + 1 < len(xs):
        right_x = xs[ci + 1]
        a, b = get_tile(xs[ci], ys[ri]), get_tile(right_x, ys[ri])
        if a is not None and b is not None:
            out['h_shift'] = calculate_horizontal_shift(a, b, mx, acq.dtype, normalization)
    if ri + 1 < len(ys):
        bottom_y = ys[ri + 1]
        a, b = get_tile(xs[ci], ys[ri]), get_tile(xs[ci], bottom_y)
        if a is not None and b is not None:
            out['v_shift'] = calculate_vertical_shift(a, b, my, acq.dtype, normalization)
    if scan_pattern == 'S-Pattern':
        out['h_shift_rev'] = (0, 0)
        out['h_shift_rev_odd'] = 0
        if right_x and bottom_y:
            a, b = get_tile(xs[ci], bottom_y), get_tile(right_x, bottom_y)
            if a is not None and b is not None:
                out['h_shift_rev'] = calculate_horizontal_shift(a, b, mx, acq.dtype, normalization)
                out['h_shift_rev_odd'] = ri % 2 == 0
    return out


def stitch_region(acq: Acquisition, t, region, read_image, use_registration=False, shifts=None,
                  flatfields: Optional[Dict[int, np.ndarray]] = None, apply_flat=False) -> np.ndarray:
    """Overwrite fusion of one (t, region) into a (1, C, Z, Hc, Wc) canvas
    (stitcher.py:639-689 + :544-611)."""
    shifts = shifts or {}
    plan = plan_region(acq, t, region, use_registration, shifts.get('h_shift', (0, 0)),
                       shifts.get('v_shift', (0, 0)), shifts.get('h_shift_rev'),
                       shifts.get('h_shift_rev_odd', 0), 1)
    canvas = np.zeros((1, len(acq.monochrome_channels), acq.num_z, plan.canvas_h, plan.canvas_w), dtype=acq.dtype)
    for f in plan.files:
        tile = read_image(f['filepath'])
        if f['rgb'] >= 0:
            tile = tile[:, :, f['rgb']]
        if apply_flat:
            tile = apply_flatfield(tile, (flatfields or {}).get(f['c']), acq.dtype)
        place_tile(canvas, f['c'], f['z'], tile, f['x_px'], f['y_px'], f['top'], f['bottom'], f['left'], f['right'])
    return canvas


# --------------------------------------------------------------------------------------
# array-level fusion (what the device kernel is checked against on synthetic stacks)
# --------------------------------------------------------------------------------------


def clip_rects(rects: np.ndarray, canvas_h: int, canvas_w: int) -> np.ndarray:
    """rects[n,6] = (src_y0, src_x0, h, w, dst_y, dst_x) -> the same after the canvas clip of
    stitcher.py:590-594 (python slice semantics; dst offsets are >= 0 on this path)."""
    out = rects.copy()
    out[:, 2] = np.maximum(0, np.minimum(rects[:, 2], canvas_h - rects[:, 4]))
    out[:, 3] = np.maximum(0, np.minimum(rects[:, 3], canvas_w - rects[:, 5]))
    return out


def fuse_plane_overwrite(tiles, rects: np.ndarray, canvas_h: int, canvas_w: int,
                         flatfield: Optional[np.ndarray] = None) -> np.ndarray:
    """Last-writer-wins fusion of one (c, z) plane.  ``tiles[i]`` is the i-th tile in
    write order, ``rects[i]`` its (src_y0, src_x0, h, w, dst_y, dst_x)."""
    dtype = tiles[0].dtype
    out = np.zeros((canvas_h, canvas_w), dtype=dtype)
    for tile, (sy, sx, h, w, dy, dx) in zip(tiles, clip_rects(np.asarray(rects), canvas_h, canvas_w)):
        if h <= 0 or w <= 0:
            continue
        src = tile[sy:sy + h, sx:sx + w]
        if flatfield is not None:
            src = apply_flatfield(src, flatfield[sy:sy + h, sx:sx + w], dtype.type)
        out[dy:dy + h, dx:dx + w] = src
    return out


def feather_weight(h: int, w: int) -> np.ndarray:
    """Distance-to-nearest-edge weight of a full tile, 1 at the border (float32)."""
    wy = np.minimum(np.arange(h) + 1, h - np.arange(h)).astype(np.float32)
    wx = np.minimum(np.arange(w) + 1, w - np.arange(w)).astype(np.float32)
    return np.minimum(wy[:, None], wx[None, :])


def fuse_plane_feather(tiles, rects: np.ndarray, canvas_h: int, canvas_w: int,
                       flatfield: Optional[np.ndarray] = None, out_dtype=np.float32) -> np.ndarray:
    """EXTENSION (not in the reference): distance-weighted blend.  Where ONE tile covers a voxel
    the output is that tile's value; where several do, out = sum_i w_i v_i / sum_i w_i in float32,
    tiles accumulated in write order.  The flatfield divide (float32, no clip) is applied to v_i
    first.  Integer outputs round to nearest-even and clip.  Uncovered voxels are 0."""
    tile_h, tile_w = tiles[0].shape
    wfull = feather_weight(tile_h, tile_w)
    acc = np.zeros((canvas_h, canvas_w), dtype=np.float32)
    wsum = np.zeros((canvas_h, canvas_w), dtype=np.float32)
    last = np.zeros((canvas_h, canvas_w), dtype=np.float32)
    count = np.zeros((canvas_h, canvas_w), dtype=np.int32)
    for tile, (sy, sx, h, w, dy, dx) in zip(tiles, clip_rects(np.asarray(rects), canvas_h, canvas_w)):
        if h <= 0 or w <= 0:
            continue
        v = tile[sy:sy + h, sx:sx + w].astype(np.float32)
        if flatfield is not None:
            with np.errstate(all='ignore'):
                v = v / flatfield[sy:sy + h, sx:sx + w].astype(np.float32)
        wt = wfull[sy:sy + h, sx:sx + w]
        acc[dy:dy + h, dx:dx + w] += wt * v
        wsum[dy:dy + h, dx:dx + w] += wt
        last[dy:dy + h, dx:dx + w] = v
        count[dy:dy + h, dx:dx + w] += 1
    with np.errstate(all='ignore'):
        out = np.where(count > 1, acc / wsum, np.where(count == 1, last, np.float32(0))).astype(np.float32)
    if np.issubdtype(np.dtype(out_dtype), np.integer):
        info = np.iinfo(out_dtype)
        return np.clip(np.rint(out), info.min, info.max).astype(out_dtype)
    return out


# --------------------------------------------------------------------------- pyramid (8f row 2)
def pyramid_nearest(image: np.ndarray, num_levels: int) -> List[np.ndarray]:
    """The multiscale levels the reference stores: ``ome_zarr.scale.Scaler(max_layer=num_levels - 1)
    .nearest(image)`` (stitcher.py:797-798).  ome_zarr is a third-party dependency absent from
    /root/reference and from this image (unpinned in install_requirements.sh:55); its published
    algorithm: each level is, plane by plane over the last two axes,
    ``skimage.transform.resize(plane, (Y // 2, X // 2), order=0, preserve_range=True,
    anti_aliasing=False).astype(dtype)`` of the level before.  Order-0 resize samples source coordinate
    ``(o + 0.5) * in / out - 0.5`` rounded half up; with ``out = in // 2`` that is index ``2 * o + 1``
    for every o, even or odd ``in`` (the fractional part lies in [1/(2 out), 1 - 1/(2 out)], never on a
    tie for odd sizes; exactly on the tie .5 -> rounds up for even sizes).

    Pinned by tests/golden/pyramid_vectors.npz: scikit-image 0.18.3's resize (odd sizes; for an exactly
    halved dimension 0.18.3 resolves the tie by rounding noise, which is not a behaviour to reproduce)
    and scipy.ndimage.zoom(order=0, grid_mode=True), the path scikit-image >= 0.19 takes (all sizes)."""
    levels = [np.asarray(image)]
    for _ in range(1, max(1, int(num_levels))):
        prev = levels[-1]
        y, x = prev.shape[-2:]
        if y < 2 or x < 2:
            break
        levels.append(np.ascontiguousarray(prev[..., 1::2, 1::2][..., :y // 2, :x // 2]))
    return levels

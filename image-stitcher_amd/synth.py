"""Seeded synthetic Squid acquisitions.

One counter-based generator, defined in integer arithmetic only, so the same tiles
come out of numpy here (any numpy version, either interpreter) and out of the HIP
generator ``sq_synth_tiles`` on the device (csrc/synth.hip) bit for bit:

    mix(x)      = splitmix64 finaliser
    h(s, Y, X)  = mix(mix(s + G*Y) ^ (K*X))                    (uint64, wrapping)
    scene(Y, X) = 1000 + sum_{dy,dx in {0,1}} (h(s, Y+dy, X+dx) >> 33) % 10000
    tile[y, x]  = scene(oy + y, ox + x) + ((h(n, y, x) >> 33) % (2*noise+1)) - noise

Tile (r, c) of a grid looks at the scene from origin
``(base + r*(H-ov_y) + c*jy, base + c*(W-ov_x) + r*jx)``: the stage steps by the
nominal pitch written to coordinates.csv, plus a planted drift (jy, jx) that
registration has to find, i.e. truth is h_shift = (jy, -ov_x), v_shift = (-ov_y, jx).

Folder layout written by ``write_acquisition`` is what the reference parses
(stitcher.py:121-201): ``acquisition parameters.json``, ``<t>/coordinates.csv``,
``<t>/<region>_<fov>_<z>_<channel>.tiff``.
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .tiffio import write_tiff

_G = np.uint64(0x9E3779B97F4A7C15)
_K = np.uint64(0xD1B54A32D192ED03)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_NOISE_SALT = 0x5851F42D4C957F2D
_MASK = (1 << 64) - 1

DEFAULT_CHANNELS = ('Fluorescence 405 nm Ex', 'Fluorescence 488 nm Ex',
                    'Fluorescence 561 nm Ex', 'Fluorescence 638 nm Ex')


def _mix(x: np.ndarray) -> np.ndarray:
    x = x ^ (x >> np.uint64(30))
    x = x * _M1
    x = x ^ (x >> np.uint64(27))
    x = x * _M2
    return x ^ (x >> np.uint64(31))


def hash2d(seed: int, yy: np.ndarray, xx: np.ndarray) -> np.ndarray:
    """h(seed, Y, X) for int64 coordinate arrays (broadcast), uint64 result."""
    with np.errstate(over='ignore'):
        s = np.uint64(seed & _MASK)
        a = _mix(s + _G * yy.astype(np.int64).astype(np.uint64))
        return _mix(a ^ (_K * xx.astype(np.int64).astype(np.uint64)))


def scene_patch(seed: int, y0: int, x0: int, h: int, w: int) -> np.ndarray:
    """scene(Y, X) for Y in [y0, y0+h), X in [x0, x0+w) as int64."""
    yy = (np.arange(h + 1, dtype=np.int64) + y0)[:, None]
    xx = (np.arange(w + 1, dtype=np.int64) + x0)[None, :]
    v = ((hash2d(seed, yy, xx) >> np.uint64(33)) % np.uint64(10000)).astype(np.int64)
    return 1000 + v[:-1, :-1] + v[1:, :-1] + v[:-1, 1:] + v[1:, 1:]


def noise_patch(seed: int, h: int, w: int, amp: int) -> np.ndarray:
    if amp <= 0:
        return np.zeros((h, w), dtype=np.int64)
    yy = np.arange(h, dtype=np.int64)[:, None]
    xx = np.arange(w, dtype=np.int64)[None, :]
    v = (hash2d(seed, yy, xx) >> np.uint64(33)) % np.uint64(2 * amp + 1)
    return v.astype(np.int64) - amp


@dataclasses.dataclass
class GridSpec:
    """One synthetic acquisition: regions x timepoints x (rows x cols) x nz x channels."""
    rows: int
    cols: int
    tile_h: int
    tile_w: int
    ov_y: int
    ov_x: int
    jy: int = 3          # y drift per column step (h_shift[0])
    jx: int = -2         # x drift per row step    (v_shift[1])
    channels: Sequence[str] = (DEFAULT_CHANNELS[1],)
    nz: int = 1
    nt: int = 1
    regions: Sequence[str] = ('R0',)
    seed: int = 0
    dtype: str = 'uint16'
    pixel_binning: int = 2
    scan_pattern: str = 'Unidirectional'
    rev_ov_x: Optional[int] = None   # S-Pattern: overlap of reversed rows (default ov_x)
    rev_jy: Optional[int] = None     # S-Pattern: y drift per column step on reversed rows
    noise: int = 200
    base: int = 4096
    rgb_channels: Sequence[str] = ()   # channels written as H x W x 3 uint8 files (dtype must be uint8)
    missing: Sequence[Tuple[int, int, int, int]] = ()   # (fov, z, channel index, t) files NOT written (ragged input)
    region_dims: Sequence[Tuple[str, int, int]] = ()   # (region, rows, cols) for regions whose grid is not rows x cols
    blank_fovs: Sequence[int] = ()   # these FOVs are written as a constant image (cap on, empty well) in every plane
    stage_jitter_um: float = 0.0     # coordinate-only tests: every FOV's stage position is off the grid by up to this
    sensor_pixel_size_um: float = 5.0
    magnification: float = 10.0
    tube_lens_mm: float = 180.0
    dz_um: float = 1.5

    @property
    def pixel_size_um(self) -> float:
        # same arithmetic as the reference's get_pixel_size (stitcher.py:131-139)
        focal = self.tube_lens_mm / self.magnification
        return self.sensor_pixel_size_um / (self.tube_lens_mm / focal)

    @property
    def n_tiles(self) -> int:
        return self.rows * self.cols

    def dims_of(self, region: str) -> Tuple[int, int]:
        """(rows, cols) of one region's grid (plates whose edge wells hold fewer FOVs)."""
        for name, rows, cols in self.region_dims:
            if name == region:
                return int(rows), int(cols)
        return self.rows, self.cols

    def row_reversed(self, r: int) -> bool:
        return self.scan_pattern == 'S-Pattern' and (r % 2 == 1)

    def fov_index(self, r: int, c: int, cols: Optional[int] = None) -> int:
        cols = self.cols if cols is None else cols
        return r * cols + (cols - 1 - c if self.row_reversed(r) else c)

    def origin(self, r: int, c: int) -> Tuple[int, int]:
        ov_x, jy = self.ov_x, self.jy
        if self.row_reversed(r):
            ov_x = self.ov_x if self.rev_ov_x is None else self.rev_ov_x
            jy = self.jy if self.rev_jy is None else self.rev_jy
        oy = self.base + r * (self.tile_h - self.ov_y) + c * jy
        ox = self.base + c * (self.tile_w - ov_x) + r * self.jx
        return oy, ox

    def stage_mm(self, r: int, c: int) -> Tuple[float, float]:
        """Nominal stage position written to coordinates.csv (no drift)."""
        px = self.pixel_size_um
        x_mm = 10.0 + c * (self.tile_w - self.ov_x) * px / 1000
        y_mm = 20.0 + r * (self.tile_h - self.ov_y) * px / 1000
        if self.stage_jitter_um:   # deterministic, different for every FOV and axis
            h = int(hash2d(self.seed + 77, np.array([r]), np.array([c]))[0])
            x_mm += ((h & 0xFFFF) / 65535.0 - 0.5) * 2 * self.stage_jitter_um / 1000
            y_mm += (((h >> 16) & 0xFFFF) / 65535.0 - 0.5) * 2 * self.stage_jitter_um / 1000
        return x_mm, y_mm

    def scene_seed(self, region_idx: int, t: int, z: int, ch: int) -> int:
        return (self.seed * 1000003 + region_idx * 100 + t) * 1000003 + z * 7919 + ch * 104729

    def noise_seed(self, region_idx: int, t: int, z: int, ch: int, fov: int) -> int:
        return (self.scene_seed(region_idx, t, z, ch) ^ _NOISE_SALT) + 31 * fov + 1

    def tile(self, r: int, c: int, region_idx: int = 0, t: int = 0, z: int = 0, ch: int = 0) -> np.ndarray:
        oy, ox = self.origin(r, c)
        v = scene_patch(self.scene_seed(region_idx, t, z, ch), oy, ox, self.tile_h, self.tile_w)
        v = v + noise_patch(self.noise_seed(region_idx, t, z, ch, self.fov_index(r, c)),
                            self.tile_h, self.tile_w, self.noise)
        if self.dtype == 'uint8':
            return (v >> 8).astype(np.uint8)
        return v.astype(np.uint16)

    def tile_stack(self, region_idx: int = 0, t: int = 0, z: int = 0, ch: int = 0) -> np.ndarray:
        """[rows*cols, H, W] in row-major (r, c) order."""
        out = np.empty((self.n_tiles, self.tile_h, self.tile_w), dtype=self.dtype)
        for r in range(self.rows):
            for c in range(self.cols):
                out[r * self.cols + c] = self.tile(r, c, region_idx, t, z, ch)
        return out

    def acquisition_parameters(self) -> dict:
        return {
            'objective': {'magnification': self.magnification, 'tube_lens_f_mm': self.tube_lens_mm},
            'sensor_pixel_size_um': self.sensor_pixel_size_um,
            'tube_lens_mm': self.tube_lens_mm,
            'pixel_binning': self.pixel_binning,
            'dz(um)': self.dz_um,
        }


def synthetic_flatfield(h: int, w: int, dtype=np.float32) -> np.ndarray:
    """Smooth 0.8 + 0.4*(separable raised cosine) illumination profile."""
    wy = 0.5 - 0.5 * np.cos(2 * np.pi * (np.arange(h, dtype=np.float64) + 0.5) / h)
    wx = 0.5 - 0.5 * np.cos(2 * np.pi * (np.arange(w, dtype=np.float64) + 0.5) / w)
    return (0.8 + 0.4 * wy[:, None] * wx[None, :]).astype(dtype)


def channel_file_token(channel: str) -> str:
    return channel.replace(' ', '_')


def write_acquisition(spec: GridSpec, root: str, ext: str = 'tiff') -> List[str]:
    """Materialise ``spec`` as a Squid acquisition folder; returns the tile paths.  ``ext``: 'tiff' (own
    writer) or 'bmp' / 'png' (uint8 tiles, through PIL) -- Squid saves .bmp or .tiff."""
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, 'acquisition parameters.json'), 'w') as fh:
        json.dump(spec.acquisition_parameters(), fh, indent=2)
    paths = []
    for t in range(spec.nt):
        tdir = os.path.join(root, str(t))
        os.makedirs(tdir, exist_ok=True)
        lines = ['region,fov,z_level,x (mm),y (mm),z (um)']
        for ri, region in enumerate(spec.regions):
            rows, cols = spec.dims_of(region)
            for r in range(rows):
                for c in range(cols):
                    fov = spec.fov_index(r, c, cols)
                    x_mm, y_mm = spec.stage_mm(r, c)
                    for z in range(spec.nz):
                        lines.append(f'{region},{fov},{z},{x_mm!r},{y_mm!r},{z * spec.dz_um!r}')
                        for ci, ch in enumerate(spec.channels):
                            if (fov, z, ci, t) in set(map(tuple, spec.missing)):
                                continue
                            p = os.path.join(tdir, f'{region}_{fov}_{z}_{channel_file_token(ch)}.{ext}')
                            img = spec.tile(r, c, ri, t, z, ci)
                            if fov in spec.blank_fovs:
                                img = np.full_like(img, 500 % (int(np.iinfo(img.dtype).max) + 1))
                            if ch in spec.rgb_channels:   # three different planes of the same scene family
                                img = np.stack([img, spec.tile(r, c, ri, t, z, ci + 17), spec.tile(r, c, ri, t, z, ci + 31)], axis=-1)
                            if ext in ('tiff', 'tif'):
                                write_tiff(p, img)
                            else:
                                from PIL import Image
                                if img.dtype != np.uint8:
                                    raise ValueError(f".{ext} tiles must be uint8")
                                Image.fromarray(img).save(p)
                            paths.append(p)
        with open(os.path.join(tdir, 'coordinates.csv'), 'w') as fh:
            fh.write('\n'.join(lines) + '\n')
    return paths


def write_acquisition_device(spec: GridSpec, root: str, device, workers: int = 16) -> List[str]:
    """``write_acquisition`` for big grids: tiles come from the device generator (sq_synth_tiles, bit for bit
    the numpy generator above) one (t, region, z, channel) plane at a time and are written by a thread
    pool.  Monochrome channels only.  Test / tool helper: needs the GPU."""
    from concurrent.futures import ThreadPoolExecutor
    from . import native
    if spec.rgb_channels:
        raise ValueError("write_acquisition_device writes monochrome channels only")
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, 'acquisition parameters.json'), 'w') as fh:
        json.dump(spec.acquisition_parameters(), fh, indent=2)
    missing = set(map(tuple, spec.missing))
    paths = []
    with ThreadPoolExecutor(max_workers=workers) as pool:
        for t in range(spec.nt):
            tdir = os.path.join(root, str(t))
            os.makedirs(tdir, exist_ok=True)
            lines = ['region,fov,z_level,x (mm),y (mm),z (um)']
            for ri, region in enumerate(spec.regions):
                rows, cols = spec.dims_of(region)
                cells = [(r, c, spec.fov_index(r, c, cols)) for r in range(rows) for c in range(cols)]
                for r, c, fov in cells:
                    x_mm, y_mm = spec.stage_mm(r, c)
                    for z in range(spec.nz):
                        lines.append(f'{region},{fov},{z},{x_mm!r},{y_mm!r},{z * spec.dz_um!r}')
                for z in range(spec.nz):
                    for ci, ch in enumerate(spec.channels):
                        desc = np.zeros(len(cells), dtype=native.SYNTH_DTYPE)
                        for i, (r, c, fov) in enumerate(cells):
                            oy, ox = spec.origin(r, c)
                            desc[i] = (spec.scene_seed(ri, t, z, ci) % 2 ** 64, spec.noise_seed(ri, t, z, ci, fov) % 2 ** 64, oy, ox)
                        plane = native.synth_tiles(desc, spec.tile_h, spec.tile_w, spec.noise, spec.dtype, device).cpu().numpy()
                        jobs = []
                        for i, (r, c, fov) in enumerate(cells):
                            if (fov, z, ci, t) in missing:
                                continue
                            p = os.path.join(tdir, f'{region}_{fov}_{z}_{channel_file_token(ch)}.tiff')
                            jobs.append((p, plane[i]))
                            paths.append(p)
                        list(pool.map(lambda job: write_tiff(*job), jobs))
            with open(os.path.join(tdir, 'coordinates.csv'), 'w') as fh:
                fh.write('\n'.join(lines) + '\n')
    return paths

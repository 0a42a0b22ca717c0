"""``Stitcher``: the reference's class surface (stitcher.py:31) over the MI355X core.

Same constructor, same method names / argument meaning / error behaviour for the methods a
front-end touches -- ``run``, ``calculate_shifts``, ``calculate_horizontal_shift``,
``calculate_vertical_shift``, ``normalize_image``, ``apply_flatfield_correction``,
``calculate_output_dimensions``, ``stitch_region`` -- and the same state attributes
(``h_shift``, ``v_shift``, ``h_shift_rev``, ``h_shift_rev_odd``, ``flatfields``,
``acquisition_metadata``, ``x_positions``, ``y_positions``, ...).  What differs is inside:

* registration and fusion run as HIP kernels through the C-ABI (``native``); there is no CPU
  path -- a missing library or GPU raises;
* a region is fused in one launch over all its (channel, z) planes from a device-resident
  tile stack, instead of one dask ``__setitem__`` layer per file;
* metadata lookups are indexed once (the reference rescans per file).

Qt is not required: signals are plain callback lists with ``emit``/``connect``.
"""
from __future__ import annotations

import json
import math
import os
import random
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import native, placement, registration, sharding
from . import omezarr
from .omezarr import write_ome_zarr
from .ometiff import write_ome_tiff
from .placement import Shifts
from .stitcher_parameters import StitchingParameters
from .tiffio import read_image, read_image_into

_IMAGE_EXT = ('.bmp', '.tiff', 'tif', 'jpg', 'jpeg', 'png')   # as the reference spells them (stitcher.py:169)


class Signal:
    """Tiny stand-in for a Qt signal (stitcher.py:33-37)."""

    def __init__(self, *types):
        self._slots = []

    def connect(self, fn):
        self._slots.append(fn)

    def emit(self, *args):
        for fn in list(self._slots):
            fn(*args)


class Stitcher:
    def __init__(self, params: StitchingParameters, device=None, fusion_mode: str = 'overwrite',
                 normalization: Optional[str] = 'phase', zarr_compression: str = 'blosc',
                 per_region_registration: bool = False, flatfield_estimator: str = 'auto',
                 all_pairs_registration: bool = False):
        self.update_progress = Signal(int, int)
        self.getting_flatfields = Signal()
        self.starting_stitching = Signal()
        self.starting_saving = Signal(bool)
        self.finished_saving = Signal(str, object)

        self.params = params
        params.validate()
        self.input_folder = params.input_folder
        self.output_folder = params.stitched_folder
        self.output_format = params.output_format
        self.merge_timepoints = getattr(params, 'merge_timepoints', False)
        self.merge_hcs_regions = getattr(params, 'merge_hcs_regions', False)
        self.per_timepoint_region_output_template = os.path.join(
            self.output_folder, "{timepoint}_stitched", "{region}_stitched" + self.output_format)
        self.apply_flatfield = params.apply_flatfield
        self.use_registration = params.use_registration
        if self.use_registration:
            self.registration_channel = params.registration_channel
            self.registration_z_level = params.registration_z_level
            self.dynamic_registration = params.dynamic_registration   # stored and never read, like the reference (stitcher.py:92)
        self.scan_pattern = params.scan_pattern
        if fusion_mode not in ('overwrite', 'feather'):
            raise ValueError("fusion_mode must be 'overwrite' or 'feather'")
        self.fusion_mode = fusion_mode            # 'feather' is an extension the reference lacks
        self.normalization = normalization        # scikit-image >= 0.19 default is 'phase'
        if zarr_compression not in ('blosc', 'zlib', 'none'):
            raise ValueError("zarr_compression must be 'blosc', 'zlib' or 'none'")
        self.zarr_compression = zarr_compression
        if flatfield_estimator not in ('auto', 'basic', 'basicpy', 'mean'):
            raise ValueError("flatfield_estimator must be 'auto', 'basic', 'basicpy' or 'mean'")
        self.flatfield_estimator = flatfield_estimator
        self.flatfield_estimator_used = None      # set by get_flatfields
        self.flatfield_info = None
        # False: shifts are measured once, on the first timepoint and region, and used everywhere (the
        # reference, stitcher.py:1244-1246).  True: every (timepoint, region) is registered on its own
        # tiles before it is fused (BASELINE config 5: per-well registration).
        self.per_region_registration = bool(per_region_registration) and self.use_registration
        # Extension of this build (the north star's batched registration), behind a flag of its own: every adjacent pair of
        # the registration plane, per-axis median.  The reference's --dynamic-registration is parsed, stored and ignored
        # (stitcher.py:92, stitcher_parameters.py:24), so that flag keeps the centre-pair result here too.
        self.all_pairs_registration = bool(all_pairs_registration) and self.use_registration
        self.batch_bytes_limit = 4 << 30          # tile bytes staged (pinned + device) per ingest batch
        self._device = device
        self._plan_cache: Dict[tuple, native.FusePlan] = {}
        self._buffer_cache: Dict[tuple, object] = {}
        # the canvas' home: device memory mapped over all memory classes of the card (native.DeviceArena), kept for the next
        # region; canvases below canvas_arena_min_bytes (and every canvas when the platform lacks virtual memory
        # management) come from a plain allocation
        self.canvas_arena_min_bytes = int(os.environ.get('SQ_CANVAS_ARENA_MIN_BYTES', 2 << 30))
        self._arena = None
        self._arena_unsupported = False
        # the stream writer of the .ome.zarr path lives across regions of one geometry (its two slots alternate: the chunks of
        # region k are written while region k + 1 is read, registered and fused); run() drains it once at the end
        self._stream_writer = None
        self._defer_drain = False
        self.canvas_arena_info = None
        self.init_stitching_parameters()

    # ------------------------------------------------------------------ state
    def init_stitching_parameters(self):
        """(stitcher.py:98-119)"""
        self.pixel_size_um = None
        self.pixel_binning = 1
        self.acquisition_params = None
        self.timepoints = []
        self.regions = []
        self.channel_names = []
        self.monochrome_channels = []
        self.monochrome_colors = []
        self.num_z = self.num_c = self.num_t = 1
        self.input_height = self.input_width = 0
        self.num_pyramid_levels = 5
        self.flatfields = {}
        self.acquisition_metadata = {}
        self.dtype = np.uint16
        self.chunks = None
        self.h_shift = (0, 0)
        if self.scan_pattern == 'S-Pattern':
            self.h_shift_rev = (0, 0)
            self.h_shift_rev_odd = 0
        self.v_shift = (0, 0)
        self.x_positions = set()
        self.y_positions = set()
        self._region_index: Dict[tuple, Dict[tuple, dict]] = {}

    @property
    def device(self):
        import torch
        if self._device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("Stitcher needs an MI355X (torch.cuda is not available); there is no CPU path")
            self._device = torch.device('cuda', torch.cuda.current_device())
        return self._device

    # --------------------------------------------------------------- metadata
    def get_timepoints(self):
        """(stitcher.py:121-124)"""
        self.timepoints = [d for d in os.listdir(self.input_folder)
                           if os.path.isdir(os.path.join(self.input_folder, d)) and d.isdigit()]
        self.timepoints.sort(key=int)
        return self.timepoints

    def extract_acquisition_parameters(self):
        with open(os.path.join(self.input_folder, 'acquisition parameters.json'), 'r') as fh:
            self.acquisition_params = json.load(fh)

    def get_pixel_size(self):
        """(stitcher.py:131-140)"""
        ap = self.acquisition_params
        obj_focal_length_mm = ap['objective']['tube_lens_f_mm'] / ap['objective']['magnification']
        actual_mag = ap['tube_lens_mm'] / obj_focal_length_mm
        self.pixel_binning = ap.get('pixel_binning', 1)
        self.pixel_size_um = ap['sensor_pixel_size_um'] / actual_mag
        print(f"[metadata] pixel size {self.pixel_size_um} um")

    @staticmethod
    def _read_coordinates(path: str) -> Dict[tuple, Tuple[float, float, float]]:
        """coordinates.csv -> {(region, fov, z): (x mm, y mm, z um)}, first row wins like
        ``coord_row.iloc[0]`` (stitcher.py:176-186)."""
        table: Dict[tuple, Tuple[float, float, float]] = {}
        with open(path) as fh:
            header = [h.strip() for h in fh.readline().rstrip('\n').split(',')]
            col = {name: i for i, name in enumerate(header)}
            for line in fh:
                parts = line.rstrip('\n').split(',')
                if len(parts) < len(header):
                    continue
                key = (parts[col['region']], int(parts[col['fov']]), int(parts[col['z_level']]))
                if key not in table:
                    table[key] = (float(parts[col['x (mm)']]), float(parts[col['y (mm)']]), float(parts[col['z (um)']]))
        return table

    def parse_acquisition_metadata(self):
        """File names -> ``acquisition_metadata`` in sorted-filename order (stitcher.py:143-257)."""
        self.acquisition_metadata = {}
        self._region_index = {}
        regions, channels = set(), set()
        max_z = max_fov = 0
        for timepoint in self.timepoints:
            image_folder = os.path.join(self.input_folder, str(timepoint))
            print(f"[metadata] timepoint {timepoint}: {image_folder}")
            try:
                coords = self._read_coordinates(os.path.join(image_folder, 'coordinates.csv'))
            except FileNotFoundError:
                print(f"Warning: timepoint {timepoint} has no coordinates.csv, skipped")
                continue
            files = sorted(f for f in os.listdir(image_folder) if f.endswith(_IMAGE_EXT) and 'focus_camera' not in f)
            for file in files:
                parts = file.split('_', 3)
                region, fov, z_level = parts[0], int(parts[1]), int(parts[2])
                channel = os.path.splitext(parts[3])[0].replace("_", " ").replace("full ", "full_")
                pos = coords.get((region, fov, z_level))
                if pos is None:
                    print(f"Warning: {file} has no row in coordinates.csv, skipped")
                    continue
                key = (int(timepoint), region, fov, z_level, channel)
                rec = {'filepath': os.path.join(image_folder, file), 'x': pos[0], 'y': pos[1], 'z': pos[2],
                       'channel': channel, 'z_level': z_level, 'region': region, 'fov_idx': fov, 't': int(timepoint)}
                self.acquisition_metadata[key] = rec
                self._region_index.setdefault((int(timepoint), region), {})[key] = rec
                regions.add(region)
                channels.add(channel)
                max_z = max(max_z, z_level)
                max_fov = max(max_fov, fov)
        self.regions = sorted(regions)
        self.channel_names = sorted(channels)
        self.num_t = len(self.timepoints)
        self.num_z = max_z + 1
        self.num_fovs_per_region = max_fov + 1
        if not self.acquisition_metadata:
            raise ValueError(f"No image files with coordinates found under {self.input_folder}")
        first_key = next(iter(self.acquisition_metadata))
        first = self.acquisition_metadata[first_key]
        first_image = read_image(first['filepath'])
        self.dtype = first_image.dtype.type
        if first_image.ndim in (2, 3):
            self.input_height, self.input_width = first_image.shape[:2]
        else:
            raise ValueError(f"Unexpected image shape: {first_image.shape}")
        self.chunks = (1, 1, 1, 512, 512)
        self.monochrome_channels = []
        for channel in self.channel_names:
            ck = (first['t'], first['region'], first['fov_idx'], first['z_level'], channel)
            img = read_image(self.acquisition_metadata[ck]['filepath'])
            if img.ndim == 3 and img.shape[2] == 3:
                base = channel.split('_')[0]
                self.monochrome_channels.extend([f"{base}_R", f"{base}_G", f"{base}_B"])
            else:
                self.monochrome_channels.append(channel)
        self.num_c = len(self.monochrome_channels)
        self.monochrome_colors = [self.get_channel_color(n) for n in self.monochrome_channels]
        print(f"[metadata] regions {self.regions}; channels {self.channel_names}")
        print(f"[metadata] tile {self.input_height} x {self.input_width} {np.dtype(self.dtype)}")
        print(f"[metadata] {self.num_z} z levels, {self.num_t} timepoints, {self.num_fovs_per_region} fovs per region")
        print(f"[metadata] {self.num_c} output channels: {self.monochrome_channels}")

    def get_region_data(self, t, region):
        """(stitcher.py:260-280) -- served from an index built once instead of a full scan."""
        data = self._region_index.get((int(t), region))
        if not data:
            raise ValueError(f"No data found for timepoint {int(t)}, region {region}")
        return data

    def get_channel_color(self, channel_name):
        """(stitcher.py:282-296)"""
        for key, color in (('405', 0x0000FF), ('488', 0x00FF00), ('561', 0xFFCF00), ('638', 0xFF0000),
                           ('730', 0x770000), ('_B', 0x0000FF), ('_G', 0x00FF00), ('_R', 0xFF0000)):
            if key in channel_name:
                return color
        return 0xFFFFFF

    def get_rows_and_columns(self):
        """(stitcher.py:1220-1223)"""
        return sorted(set(r[0] for r in self.regions)), sorted(set(r[1:] for r in self.regions))

    def get_tile(self, t, region, x, y, channel, z_level):
        """(stitcher.py:526-542) -> numpy image or None."""
        for value in self.get_region_data(int(t), str(region)).values():
            if value['x'] == x and value['y'] == y and value['channel'] == channel and value['z_level'] == z_level:
                try:
                    return read_image(value['filepath'])
                except FileNotFoundError:
                    print(f"Warning: cannot open {value['filepath']}")
                    return None
        print(f"Warning: region {region} has no tile at ({x}, {y}) mm for channel {channel}, z {z_level}")
        return None

    # ------------------------------------------------------------- geometry
    def _shifts(self) -> Shifts:
        rev = getattr(self, 'h_shift_rev', None) if self.scan_pattern == 'S-Pattern' else None
        return Shifts(tuple(self.h_shift), tuple(self.v_shift), None if rev is None else tuple(rev),
                      int(getattr(self, 'h_shift_rev_odd', 0)))

    def _apply_shifts(self, s: Optional[Shifts]) -> None:
        if s is None:
            return
        self.h_shift, self.v_shift = tuple(s.h_shift), tuple(s.v_shift)
        if s.h_shift_rev is not None:
            self.h_shift_rev, self.h_shift_rev_odd = tuple(s.h_shift_rev), s.h_shift_rev_odd

    def calculate_output_dimensions(self, timepoint, region):
        """(width_pixels, height_pixels); also sets x/y_positions and num_pyramid_levels
        (stitcher.py:298-354)."""
        region_data = self.get_region_data(int(timepoint), region)
        self.x_positions = sorted(set(v['x'] for v in region_data.values()))
        self.y_positions = sorted(set(v['y'] for v in region_data.values()))
        width_pixels, height_pixels = placement.canvas_size(
            len(self.x_positions), len(self.y_positions), self.input_width, self.input_height,
            use_registration=self.use_registration, shifts=self._shifts(),
            xs=self.x_positions, ys=self.y_positions, pixel_size_um=self.pixel_size_um)
        max_dimension = 1
        if len(self.regions) > 1:
            rows, columns = self.get_rows_and_columns()
            max_dimension = max(len(rows), len(columns))
        self.num_pyramid_levels = placement.pyramid_levels(width_pixels, height_pixels, max_dimension)
        return width_pixels, height_pixels

    # ---------------------------------------------------------- registration
    def normalize_image(self, img):
        """(stitcher.py:613-617) on the device: full-tile min/max, then the float64 stretch and the
        truncating cast.  ``calculate_*_shift`` never call this: the registration kernels fuse the same
        arithmetic into their first pass."""
        import torch
        img = np.ascontiguousarray(img)
        if img.ndim != 2 or img.dtype not in (np.uint8, np.uint16):
            raise ValueError(f"normalize_image takes a 2-D uint8/uint16 image, got {img.dtype} {img.shape}")
        tiles = torch.from_numpy(img[None]).to(self.device)
        return native.normalize_tiles(tiles)[0].cpu().numpy()

    def _register_two(self, img_a, img_b, max_overlap, vertical: bool):
        import torch
        a, b = np.asarray(img_a), np.asarray(img_b)
        if a.ndim != 2 or b.ndim != 2 or a.shape != b.shape:
            raise ValueError("images must be same shape")
        tiles = torch.from_numpy(np.ascontiguousarray(np.stack([a, b]))).to(self.device)
        h, w = a.shape
        make = registration.vertical_pair if vertical else registration.horizontal_pair
        pair, n0, n1 = make(0, 1, h, w, int(max_overlap))
        s, _, _ = registration.register_pairs(tiles, np.array([pair], dtype=native.PAIR_DTYPE), n0, n1, 10,
                                              self.normalization)
        return s[0], n0, n1

    def calculate_horizontal_shift(self, img_left, img_right, max_overlap):
        """(stitcher.py:500-511) -> (dy, dx) python ints."""
        s, n0, n1 = self._register_two(img_left, img_right, max_overlap, vertical=False)
        return registration.horizontal_shift_from(s, n1)

    def calculate_vertical_shift(self, img_top, img_bot, max_overlap):
        """(stitcher.py:513-524)"""
        s, n0, n1 = self._register_two(img_top, img_bot, max_overlap, vertical=True)
        return registration.vertical_shift_from(s, n0)

    def calculate_shifts(self, t, region):
        """Centre-pair registration (stitcher.py:422-498); sets h_shift / v_shift
        [/ h_shift_rev / h_shift_rev_odd]."""
        region_data = self.get_region_data(t, region)
        x_positions = sorted(set(v['x'] for v in region_data.values()))
        y_positions = sorted(set(v['y'] for v in region_data.values()))
        self.h_shift = (0, 0)
        self.v_shift = (0, 0)
        if not self.registration_channel:
            self.registration_channel = self.channel_names[0]
        elif self.registration_channel not in self.channel_names:
            print(f"Warning: Specified registration channel '{self.registration_channel}' not found. "
                  f"Using {self.channel_names[0]}.")
            self.registration_channel = self.channel_names[0]
        self.calculate_output_dimensions(int(t), region)
        max_x_overlap, max_y_overlap = placement.registration_crop_widths(
            sorted(self.x_positions), sorted(self.y_positions), self.input_width, self.input_height,
            self.pixel_size_um, self.pixel_binning)
        print(f"[registration] crop widths from the stage pitch: {max_x_overlap} px horizontal, {max_y_overlap} px vertical")
        registration.check_crop_lengths(self.input_height, self.input_width, max_x_overlap, max_y_overlap)
        if self.all_pairs_registration:
            # --all-pairs-registration (an addition of this build; NOT the reference's --dynamic-registration, which it
            # parses and never reads): every adjacent pair of the registration plane, one batch per direction, per-axis
            # median of the integer shifts
            self._calculate_shifts_all_pairs(t, region, x_positions, y_positions, max_x_overlap, max_y_overlap)
            print(f"[registration] all pairs: h_shift = {self.h_shift}, v_shift = {self.v_shift}")
            return
        cx, cy = (len(x_positions) - 1) // 2, (len(y_positions) - 1) // 2
        center_x, center_y = x_positions[cx], y_positions[cy]
        right_x = bottom_y = None
        get = lambda x, y: self.get_tile(t, region, x, y, self.registration_channel, self.registration_z_level)
        if cx + 1 < len(x_positions):
            right_x = x_positions[cx + 1]
            a, b = get(center_x, center_y), get(right_x, center_y)
            if a is not None and b is not None:
                self.h_shift = self.calculate_horizontal_shift(a, b, max_x_overlap)
            else:
                print(f"Warning: region {region}: centre or right tile missing, h_shift stays {self.h_shift}")
        if cy + 1 < len(y_positions):
            bottom_y = y_positions[cy + 1]
            a, b = get(center_x, center_y), get(center_x, bottom_y)
            if a is not None and b is not None:
                self.v_shift = self.calculate_vertical_shift(a, b, max_y_overlap)
            else:
                print(f"Warning: region {region}: centre or bottom tile missing, v_shift stays {self.v_shift}")
        if self.scan_pattern == 'S-Pattern' and right_x and bottom_y:
            a, b = get(center_x, bottom_y), get(right_x, bottom_y)
            if a is not None and b is not None:
                self.h_shift_rev = self.calculate_horizontal_shift(a, b, max_x_overlap)
                self.h_shift_rev_odd = cy % 2 == 0
                print(f"[registration] reversed rows: h_shift_rev = {self.h_shift_rev}")
            else:
                print(f"Warning: region {region}: tiles of the reversed row missing, h_shift_rev stays {self.h_shift_rev}")
        print(f"[registration] h_shift = {self.h_shift}, v_shift = {self.v_shift}")

    def _calculate_shifts_all_pairs(self, t, region, xs, ys, max_x_overlap, max_y_overlap):
        """Extension: registration over ALL adjacent tile pairs of the registration plane (batched on the
        device), reduced to the reference's state (h_shift, v_shift[, h_shift_rev]) by a per-axis median --
        a single bad tile (dust, empty field) then cannot derail the whole mosaic.

        Under ``run()`` with several ranks (``self._pair_ranks`` = (rank, world)) the pairs are SHARDED: every rank
        takes one contiguous run of the pair list (tile-row order), reads and uploads only the tiles its pairs touch,
        and the [n_pairs, 3] float64 table {dy, dx, err} is all-gathered (registration.register_all_pairs_sharded);
        the medians are host arithmetic on that table, so every rank sets the same integers."""
        import torch
        n_rows, n_cols = len(ys), len(xs)
        at = {}
        for v in self.get_region_data(t, region).values():
            if v['channel'] == self.registration_channel and v['z_level'] == self.registration_z_level:
                at[(ys.index(v['y']), xs.index(v['x']))] = v['filepath']
        if not at:
            return
        rank, world = getattr(self, '_pair_ranks', None) or (0, 1)
        pairs = registration.grid_pair_list(n_rows, n_cols, present=at)

        def load_cells(cells):
            def one(cell):
                img = read_image(at[cell])
                return img if img.ndim == 2 else img[..., 0]
            with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:
                images = list(pool.map(one, cells))
            return torch.from_numpy(np.ascontiguousarray(np.stack(images))).to(self.device)

        table = registration.register_all_pairs_sharded(
            pairs, load_cells, self.input_height, self.input_width, max_x_overlap, max_y_overlap, self.normalization,
            rank=rank, world=world, device=sharding.collective_device(self))
        self.pair_table = table     # [n_pairs, {dy, dx, err}] float64, pair order = registration.grid_pair_list
        med = registration.pair_table_medians(pairs, table, self.input_height, self.input_width, max_x_overlap,
                                              max_y_overlap, n_rows, self.scan_pattern)
        for name, value in med.items():
            setattr(self, name, bool(value) if name == 'h_shift_rev_odd' else tuple(value))

    # -------------------------------------------------------------- flatfield
    def apply_flatfield_correction(self, tile, channel_idx):
        """(stitcher.py:607-611) on one host tile through the device kernel."""
        if channel_idx not in self.flatfields:
            return tile
        import torch
        tile = np.ascontiguousarray(tile)
        h, w = tile.shape
        plan = self._plan_for(np.array([(0, 0, h, w, 0, 0)]), h, w, h, w, native.SQ_FUSE_OVERWRITE)
        canvas = torch.empty((1, h, w), dtype=native.torch_dtype_of(tile.dtype), device=self.device)
        flat = torch.from_numpy(np.ascontiguousarray(self.flatfields[channel_idx])).to(self.device)
        native.fuse_planes(plan, torch.from_numpy(tile[None, None]).to(self.device), canvas, [flat])
        return canvas[0].cpu().numpy()

    def get_flatfields(self, progress_callback=None):
        """One gain image per monochrome channel from the reference's sample of tiles -- at most 32 randomly chosen
        per timepoint, stopping once more than 48 are collected (stitcher.py:365-419).

        The reference fits them with a third-party estimator, ``basicpy.BaSiC(get_darkfield=False,
        smoothness_flatfield=1).fit(images).flatfield`` (:374-377).  ``flatfield_estimator``:
          'auto'     basicpy when it is importable (then the very same call is made and the gains are the
                     reference's), else 'basic';
          'basic'    the device restatement of the published BaSiC fit (csrc/basic.hip, defined by
                     oracle/basic_oracle.py).  basicpy and jax are absent offline, so its parity with basicpy is
                     UNPINNED; it recovers planted gains (tests/test_basic_gpu.py);
          'basicpy'  basicpy or an ImportError;
          'mean'     mean of the sample, box-smoothed, mean 1: NOT BaSiC, only on explicit request.
        Which one ran is kept in ``flatfield_estimator_used`` and written to ``flatfield_info.json`` (and
        ``shift_table.json``) beside the output.  Either way the estimate is not on the hot path: the divide by the
        gains is (``apply_flatfield_correction``, in the fusion kernel).  Flatfields already assigned to
        ``self.flatfields`` are left untouched."""
        import torch
        want = self.flatfield_estimator
        BaSiC = None
        if want in ('auto', 'basicpy'):
            try:
                from basicpy import BaSiC
            except Exception:
                if want == 'basicpy':
                    raise ImportError("flatfield_estimator='basicpy' but basicpy cannot be imported")
        used = 'basicpy' if BaSiC is not None else ('mean' if want == 'mean' else 'basic')
        self.flatfield_estimator_used = used
        self.flatfield_info = {'estimator': used, 'channels': {}}
        if used == 'basic':
            print("[flatfield] BaSiC (LADMAP, no darkfield, smoothness 1) on the device: this build's restatement of the "
                  "published algorithm -- basicpy is not installed, parity with it is unpinned")
        elif used == 'mean':
            print("[flatfield] mean / box-smooth estimate on request: this is NOT the reference's BaSiC fit")

        def estimate(images: np.ndarray, channel_name: str):
            channel_index = self.monochrome_channels.index(channel_name)
            if channel_index in self.flatfields:
                return
            info = {}
            if used == 'basicpy':
                basic = BaSiC(get_darkfield=False, smoothness_flatfield=1)
                basic.fit(images)
                self.flatfields[channel_index] = np.asarray(basic.flatfield)
            elif used == 'basic':
                stack = torch.from_numpy(np.ascontiguousarray(images)).to(self.device)
                flat, info = native.basic_fit(stack, 1.0)
                self.flatfields[channel_index] = flat.cpu().numpy()
            else:
                acc = images.astype(np.float64).mean(axis=0)
                k = max(1, min(acc.shape) // 16)
                csum = np.cumsum(np.cumsum(np.pad(acc, ((k, k), (k, k)), mode='edge'), 0), 1)
                csum = np.pad(csum, ((1, 0), (1, 0)))
                n = 2 * k + 1
                smooth = (csum[n:, n:] - csum[:-n, n:] - csum[n:, :-n] + csum[:-n, :-n]) / (n * n)
                self.flatfields[channel_index] = (smooth / smooth.mean()).astype(np.float32)
            self.flatfield_info['channels'][channel_name] = dict(info, images=int(len(images)))
            if progress_callback:
                progress_callback(channel_index + 1, self.num_c)

        for channel in self.channel_names:
            print(f"Calculating {channel} flatfield...")
            paths = []
            for t in self.timepoints:
                at_t = [v['filepath'] for k, v in self.acquisition_metadata.items()
                        if v['channel'] == channel and k[0] == int(t)]
                if not at_t:
                    print(f"WARNING: No images found for channel {channel} at timepoint {t}")
                    continue
                random.shuffle(at_t)
                paths.extend(at_t[:min(32, len(at_t))])
                if len(paths) > 48:
                    break
            if not paths:
                print(f"WARNING: No images found for channel {channel} across all timepoints")
                continue
            images = np.array([read_image(p) for p in paths])
            if images.ndim == 4 and images.shape[1] == 1:       # (N, 1, Y, X) page stacks
                images = images[:, 0]
            if images.ndim == 3:
                estimate(images, channel)
            elif images.ndim == 4 and images.shape[-1] == 3:    # RGB files: one gain image per colour
                base = channel.split('_')[0]
                for i, color in enumerate('RGB'):
                    estimate(np.ascontiguousarray(images[..., i]), f"{base}_{color}")
            else:
                raise ValueError(f"Unexpected number of dimensions in images array: {images.ndim}")

    # ----------------------------------------------------------------- fusion
    def _keep_buffers(self, key, bufs) -> None:
        """Remember staging / slot buffers for the next region of the same geometry (page-locking host
        memory is slow); one entry per kind, so a change of geometry releases the old ones."""
        for k in [k for k in self._buffer_cache if k[0] == key[0] and k != key]:
            del self._buffer_cache[k]
        self._buffer_cache[key] = bufs

    def _plan_for(self, rects, tile_h, tile_w, canvas_h, canvas_w, mode) -> native.FusePlan:
        rects = np.asarray(rects, dtype=np.int64).reshape(-1, 6)
        key = (rects.tobytes(), tile_h, tile_w, canvas_h, canvas_w, mode)
        plan = self._plan_cache.get(key)
        if plan is None:
            if len(self._plan_cache) > 16:
                self._plan_cache.clear()
            # a large overwrite plan has its work list produced on the device (native.FusePlan, csrc/plan_expand.hip):
            # the table is the host planner's byte for byte, 1.5 instead of 5.9 ms for a 32 x 32 grid
            plan = self._plan_cache[key] = native.FusePlan(rects, tile_h, tile_w, canvas_h, canvas_w, mode,
                                                           expand_on_device=len(rects) >= 64)
        return plan

    def _new_arena(self, need: int):
        """A DeviceArena of ``need`` bytes, or None when the canvas has to come from a plain allocation: the platform lacks
        virtual memory management (remembered), or the card cannot give the slices even after PyTorch's cached blocks have been
        handed back (this call only).  Anything else raises."""
        import torch
        for attempt in range(2):
            try:
                return native.DeviceArena(need, self.device)
            except native.NativeError as exc:
                if 'virtual memory management unsupported' in str(exc):
                    print(f"[canvas] no virtual memory management on this platform ({exc}); the canvas comes from a plain allocation")
                    self._arena_unsupported = True
                    return None
                if 'out of memory' not in str(exc).lower():
                    raise
                if attempt == 0:
                    torch.cuda.empty_cache()      # blocks PyTorch's allocator holds but nobody uses
                else:
                    print(f"[canvas] the card cannot give {need / 2**30:.1f} GiB of arena slices ({exc}); the canvas comes from a plain allocation")
        return None

    def _empty_canvas(self, n_planes, hc, wc, tdtype):
        """The canvas of ``stitch_planes`` ([n_planes, hc, wc], planes on 128-byte lines, never zero-filled: stitcher.py:356-362's
        da.zeros is written by the fusion kernel).  From ``canvas_arena_min_bytes`` up it is carved from a DeviceArena --
        physical slices classified by a probe and mapped round-robin over the card's memory classes, so the kernel's
        row-segment writes run at the spread-out rate (0.71 instead of 0.61-0.69 of the HBM peak on config 3, and the same
        on every box; csrc/arena.hip) -- created before the ingest buffers, while the card has memory to choose from, and
        reused for the next region once this region's canvas has been dropped."""
        need = native.canvas_bytes(n_planes, hc, wc, tdtype)
        if need < self.canvas_arena_min_bytes or self._arena_unsupported:
            return native.empty_canvas(n_planes, hc, wc, tdtype, self.device)
        busy = self._arena is not None and self._arena.in_use()      # a caller still holds the last canvas: leave it alone
        if self._arena is None or busy or self._arena.nbytes < need:
            if self._arena is not None and not busy:
                self._arena.close()
            self._arena = self._new_arena(need)
            if self._arena is None:
                return native.empty_canvas(n_planes, hc, wc, tdtype, self.device)
            self.canvas_arena_info = self._arena.info
            print(f"[canvas] arena of {self._arena.nbytes / 2**30:.1f} GiB over {self._arena.info['n_classes']} memory classes "
                  f"{self._arena.info['class_slices']} (slices of {self._arena.info['slice_bytes'] >> 20} MiB), "
                  f"{self._arena.info['create_ms']:.0f} ms")
        self._arena.reset()
        return native.empty_canvas(n_planes, hc, wc, tdtype, self.device, arena=self._arena)

    def _tile_rect(self, tile_info):
        """sq_rect of one file: placement (stitcher.py:656-679) + crop (:570-587)."""
        if self.use_registration:
            col = self.x_positions.index(tile_info['x'])
            row = self.y_positions.index(tile_info['y'])
            self.col_index, self.row_index = col, row
            return placement.registered_rect(row, col, len(self.y_positions), len(self.x_positions),
                                             self.input_width, self.input_height, self._shifts(),
                                             crop=self.fusion_mode == 'overwrite')
        return placement.coordinate_rect(tile_info['x'], tile_info['y'], min(self.x_positions), min(self.y_positions),
                                         self.input_width, self.input_height, self.pixel_size_um)

    def init_output(self, timepoint, region):
        """Device canvas (1, C, Z, Hc, Wc) (stitcher.py:356-362).  Not zero-filled: the fusion
        kernel writes every voxel, zeros included."""
        import torch
        width, height = self.calculate_output_dimensions(timepoint, region)
        shape = (1, self.num_c, self.num_z, height, width)
        print(f"region {region} timepoint {timepoint} output array dimensions: {shape}")
        return torch.empty(shape, dtype=native.torch_dtype_of(self.dtype), device=self.device)

    def stitch_region(self, timepoint, region, progress_callback=None, device_output: bool = False):
        """Fuse one (timepoint, region) -> 5-D TCZYX array of the input dtype
        (stitcher.py:639-689).  Returns numpy (host) unless ``device_output``."""
        # canvas slots z-major ("spread"): the z planes of a channel -- which share a gain image and go through the
        # fusion kernel together -- then lie num_c planes apart in the canvas allocation instead of side by side; a
        # group of planes writes fastest when they sit in different stretches of device memory (DESIGN.md 5.1 point 8)
        planes, _ = self.stitch_planes(timepoint, region, None, progress_callback, slot_order='spread')
        shape = (1, self.num_c, self.num_z, planes.shape[-2], planes.shape[-1])
        by_cz = planes.unflatten(0, (self.num_z, self.num_c)).transpose(0, 1)   # [C, Z, Hc, Wc] view of the [Z * C] slots
        if device_output:       # a strided view: planes sit on 128-byte lines, rows are dense
            return by_cz.unsqueeze(0)
        import torch
        out = torch.empty(shape, dtype=planes.dtype)
        for c in range(self.num_c):
            for z in range(self.num_z):      # one D2H copy per plane (a strided .cpu() would first copy on the device)
                out[0, c, z].copy_(by_cz[c, z])
        return out.numpy()

    def stitch_planes(self, timepoint, region, only_planes=None, progress_callback=None, stream_to=None, row_band=None,
                      slot_order: str = 'plane'):
        """Fuse the (channel, z) planes ``only_planes`` (plane = channel * num_z + z; None = all) of one
        (timepoint, region) -> (device tensor [n, Hc, Wc], sorted plane ids).  Planes are independent,
        which is what lets several GPUs share one region.

        ``stream_to``: callable ``batch -> omezarr.PlaneStreamWriter``.  When given, no region-sized canvas
        is allocated: every batch of planes is fused into one of the writer's two slots and leaves
        for disk while the next batch is read and fused; the return value is (None, plane ids).

        ``row_band`` = (y0, y1): only these canvas rows of the planes (sharding.row_bands -- one plane shared by
        several GPUs); the canvas is then y1 - y0 rows high, tiles outside the band are not even read.

        ``slot_order``: 'plane' -- the returned tensor's i-th plane is the i-th plane id; 'spread' (all planes only) --
        plane c * num_z + z sits at slot z * num_c + c, so the planes of a channel are num_c slots apart."""
        import torch
        start_time = time.time()
        region_data = self.get_region_data(int(timepoint), region)
        width, height = self.calculate_output_dimensions(timepoint, region)
        plane_ids = sorted(set(int(p) for p in only_planes)) if only_planes is not None \
            else list(range(self.num_c * self.num_z))
        if plane_ids and (plane_ids[0] < 0 or plane_ids[-1] >= self.num_c * self.num_z):
            raise ValueError(f"plane ids must lie in [0, {self.num_c * self.num_z})")
        slot_of = {p: i for i, p in enumerate(plane_ids)}
        if slot_order == 'spread':
            if only_planes is not None or stream_to is not None:
                raise ValueError("slot_order='spread' lays out ALL planes of a region in one canvas")
            slot_of = {p: (p % self.num_z) * self.num_c + p // self.num_z for p in plane_ids}
        elif slot_order != 'plane':
            raise ValueError(f"slot_order must be 'plane' or 'spread', got {slot_order!r}")
        print(f"region {region} timepoint {timepoint} output array dimensions: "
              f"{(1, self.num_c, self.num_z, height, width)}" + ("" if only_planes is None else f", planes {plane_ids}"))
        # dense rows like the reference's array, every plane on a 128-byte line (native.empty_canvas)
        y0, y1 = (0, height) if row_band is None else (int(row_band[0]), int(row_band[1]))
        if not (0 <= y0 < y1 <= height):
            raise ValueError(f"row band {row_band} outside the {height}-row canvas")
        hc, wc = y1 - y0, width
        flat_canvas = None if stream_to is not None else \
            self._empty_canvas(len(plane_ids), hc, wc, native.torch_dtype_of(self.dtype))
        th, tw = self.input_height, self.input_width
        total_tiles = len(region_data)
        print(f"Beginning stitching of {total_tiles} tiles for region {region} timepoint {timepoint}")

        # group the files by (channel, z) plane, keeping the reference's write order inside a plane
        planes: Dict[int, List[Tuple[dict, int, tuple]]] = {}
        for key, info in region_data.items():
            _, _, fov, z_level, channel = key
            rect = self._tile_rect(info)
            placement.check_rect_fits_like_numpy(rect, height, width)
            if row_band is not None:
                rect = placement.clip_rect_to_rows(rect, y0, y1)
                if rect is None:
                    continue
            if channel in self.monochrome_channels:
                targets = [(self.monochrome_channels.index(channel) * self.num_z + z_level, -1)]
            else:   # RGB file -> three monochrome channels (stitcher.py:551-556)
                base = channel.split('_')[0]
                targets = [(self.monochrome_channels.index(f"{base}_{color}") * self.num_z + z_level, i)
                           for i, color in enumerate('RGB')]
            for p, rgb in targets:
                if p in slot_of:
                    planes.setdefault(p, []).append((info, rgb, rect))

        mode = native.SQ_FUSE_OVERWRITE if self.fusion_mode == 'overwrite' else native.SQ_FUSE_FEATHER
        # planes no file touches still have to come out as zeros
        empty = [p for p in plane_ids if p not in planes] if stream_to is None else []   # streamed: fill_value
        if empty:
            zplan = self._plan_for(np.zeros((0, 6)), th, tw, hc, wc, native.SQ_FUSE_OVERWRITE)
            for p in empty:
                native.fuse_planes(zplan, torch.empty((1, 0, th, tw), dtype=flat_canvas.dtype, device=self.device),
                                   flat_canvas[slot_of[p]:slot_of[p] + 1])
        # run() streaming region after region: copies, fusion, pyramid and encoding of a region go to a stream of their own, so
        # that the NEXT region's registration -- it reads its shifts back, i.e. waits for the stream it runs on -- does not wait for
        # them (config 5: a (well, timepoint) unit is one batch; tools/cfg5_probe.py)
        side = None
        if stream_to is not None and self._defer_drain:
            if getattr(self, '_ingest_stream', None) is None:
                self._ingest_stream = torch.cuda.Stream(device=self.device)
            side = self._ingest_stream
            side.wait_stream(torch.cuda.current_stream(self.device))      # whatever the caller enqueued (flatfields, shifts) first
        import contextlib
        stream_ctx = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()

        # batch planes that share one rectangle list (normally: all of them)
        groups: Dict[bytes, List[int]] = {}
        rect_of: Dict[bytes, np.ndarray] = {}
        for p, items in planes.items():
            r = np.array([it[2] for it in items], dtype=np.int64).reshape(-1, 6)
            groups.setdefault(r.tobytes(), []).append(p)
            rect_of[r.tobytes()] = r
        # Ingest pipeline: file decode (host threads) -> pinned staging -> async H2D -> fusion, two slots
        # deep, so the files of batch k+1 are read while batch k is copied and fused.  A batch is a
        # few planes: bounded by free HBM, by host memory and by what is sensible to pin.
        free_bytes = torch.cuda.mem_get_info(self.device)[0]
        try:
            import psutil
            host_free = psutil.virtual_memory().available
        except Exception:   # pragma: no cover
            host_free = 8 << 30
        budget = max(1, min(int(free_bytes * 0.2), int(host_free * 0.1), int(self.batch_bytes_limit)))
        tdtype = native.torch_dtype_of(self.dtype)
        processed = 0
        pool = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4))
        writer = None
        stream_ctx.__enter__()
        try:
            flats_dev = {}
            if self.apply_flatfield:
                for ci, ff in self.flatfields.items():
                    flats_dev[ci] = torch.from_numpy(np.ascontiguousarray(ff)).to(self.device)
            if stream_to is not None and groups:
                widest = max(len(rect_of[sig]) for sig in groups) * th * tw * np.dtype(self.dtype).itemsize
                writer = stream_to(max(1, min(max(len(pl) for pl in groups.values()), budget // max(1, widest))))
            for sig, plist in groups.items():
                # ascending plane ids: the canvas slots of a chunk are then consecutive and the whole chunk
                # goes out in ONE launch (region_data is in file-name order, i.e. z varies before channel)
                plist = sorted(plist)
                rects = rect_of[sig]
                n = len(rects)
                plan = self._plan_for(rects, th, tw, hc, wc, mode)
                per_plane = n * th * tw * np.dtype(self.dtype).itemsize
                batch = max(1, min(len(plist), budget // max(1, per_plane)))
                if writer is not None:
                    batch = min(batch, writer.batch)
                chunks = [plist[b0:b0 + batch] for b0 in range(0, len(plist), batch)]
                # two slots also for a single chunk when run() streams region after region (config 5: a region is one chunk):
                # the next region's files are then read while this region's copy and fusion are still under way
                pipelined = writer is not None and self._defer_drain
                n_slots = 2 if (len(chunks) > 1 or pipelined) else 1
                key = ('ingest', batch, n, th, tw, n_slots, np.dtype(self.dtype).str)
                bufs = self._buffer_cache.get(key)
                if bufs is None:   # pinned staging + device mirrors, kept for the next region of the same shape
                    bufs = ([torch.empty((batch, n, th, tw), dtype=tdtype, pin_memory=True) for _ in range(n_slots)],
                            [torch.empty((batch, n, th, tw), dtype=tdtype, device=self.device) for _ in range(n_slots)],
                            [None] * n_slots, [0])
                    self._keep_buffers(key, bufs)
                # the slots' "copy and fusion finished" events live with the buffers: another group (or the next
                # region) that gets the same cached staging must wait for the H2D copy still reading it
                staging, on_dev, done, turn = bufs
                for k, chunk in enumerate(chunks):
                    slot = turn[0] % n_slots      # (the turn goes on across calls: the next region starts on the other slot)
                    turn[0] += 1
                    if done[slot] is not None:
                        done[slot].synchronize()      # the slot's previous copy and fusion have finished
                    host = staging[slot].numpy()

                    def load(job, host=host):
                        pi, ti, (info, rgb, _) = job
                        if rgb < 0 and read_image_into(info['filepath'], host[pi, ti]):
                            return      # file -> page-locked staging in one read
                        img = read_image(info['filepath'])
                        if rgb >= 0:
                            img = img[:, :, rgb]
                        elif img.ndim == 3 and img.shape[0] == 1:
                            img = img[0]
                        if img.shape != (th, tw):
                            raise ValueError(f"Unexpected tile shape: {img.shape}")
                        host[pi, ti] = img

                    jobs = [(pi, ti, it) for pi, p in enumerate(chunk) for ti, it in enumerate(planes[p])]
                    for _ in pool.map(load, jobs):
                        processed += 1
                        if progress_callback:
                            progress_callback(processed - 1, total_tiles)
                    m = len(chunk)
                    tiles = on_dev[slot][:m]
                    tiles.copy_(staging[slot][:m], non_blocking=True)
                    flats = [flats_dev.get(p // self.num_z) for p in chunk] if self.apply_flatfield else None
                    slots = [slot_of[p] for p in chunk]
                    if writer is not None:
                        native.fuse_planes(plan, tiles, writer.acquire(m), flats)
                        writer.submit([(0, p // self.num_z, p % self.num_z) for p in chunk])
                    elif m == 1 or (slots[1] > slots[0] and all(slots[i + 1] - slots[i] == slots[1] - slots[0] for i in range(m - 1))):
                        # the chunk's canvas slots are evenly spaced (side by side, or num_c apart with slot_order
                        # 'spread'): ONE launch on the strided view
                        step = slots[1] - slots[0] if m > 1 else 1
                        native.fuse_planes(plan, tiles, flat_canvas[slots[0]:slots[0] + (m - 1) * step + 1:step], flats)
                    else:
                        for pi, sl in enumerate(slots):
                            native.fuse_planes(plan, tiles[pi:pi + 1], flat_canvas[sl:sl + 1],
                                               None if flats is None else flats[pi:pi + 1])
                    done[slot] = torch.cuda.Event()
                    done[slot].record()
        finally:
            stream_ctx.__exit__(None, None, None)
            pool.shutdown(wait=True)
            if writer is not None and not self._defer_drain:
                writer.drain()      # everything of this region is on disk when the call returns (run() defers it to its end)
        if not (writer is not None and self._defer_drain):      # (run(): the writer's events order everything; it is drained at the end)
            torch.cuda.synchronize(self.device)
        print(f"Time to stitch region {region} timepoint {timepoint}: {time.time() - start_time}")
        return flat_canvas, plane_ids

    # ------------------------------------------------------------------ output
    def _zarr_path(self, timepoint, region) -> str:
        return os.path.join(self.output_folder, f"{timepoint}_stitched", f"{region}_stitched.ome.zarr")

    def _dz_um(self) -> float:
        return float(self.acquisition_params.get('dz(um)', 1.0)) if self.acquisition_params else 1.0

    def save_region_ome_zarr(self, timepoint, region, stitched_region):
        """(stitcher.py:771-859) via the package-free writer in omezarr.py; ``stitched_region`` is a
        5-D numpy array or device tensor.  The pyramid levels (Scaler.nearest, :797-798) come from the
        device kernel either way."""
        output_path = self._zarr_path(timepoint, region)
        os.makedirs(os.path.dirname(output_path), exist_ok=True)
        write_ome_zarr(output_path, stitched_region, pixel_size_um=self.pixel_size_um, dz_um=self._dz_um(),
                       channel_names=self.monochrome_channels, channel_colors=self.monochrome_colors,
                       num_levels=self.num_pyramid_levels, chunks=self.chunks or (1, 1, 1, 512, 512),
                       name=f"{region}_t{timepoint}", compression=self.zarr_compression, device=self.device)
        return output_path

    def create_region_store(self, timepoint, region):
        """Metadata of the region's OME-Zarr store (no chunks) -> (path, level shapes)."""
        output_path = self._zarr_path(timepoint, region)
        os.makedirs(os.path.dirname(output_path), exist_ok=True)
        width, height = self.calculate_output_dimensions(timepoint, region)
        shapes = omezarr.create_store(output_path, (1, self.num_c, self.num_z, height, width), self.dtype,
                                      pixel_size_um=self.pixel_size_um, dz_um=self._dz_um(),
                                      channel_names=self.monochrome_channels, channel_colors=self.monochrome_colors,
                                      num_levels=self.num_pyramid_levels, chunks=self.chunks or (1, 1, 1, 512, 512),
                                      name=f"{region}_t{timepoint}", compression=self.zarr_compression)
        return output_path, shapes

    def stream_region_to_zarr(self, timepoint, region, only_planes=None, progress_callback=None, create: bool = True,
                              row_band=None):
        """stitch_region + save_region_ome_zarr without the region ever existing in one piece: planes
        are fused a batch at a time and stream through pyramid kernel, pinned D2H copy and compression
        threads while the next batch is read and fused.  Same store as
        ``save_region_ome_zarr(t, r, stitch_region(t, r))``."""
        if create:
            output_path, shapes = self.create_region_store(timepoint, region)
        else:
            output_path = self._zarr_path(timepoint, region)
            width, height = self.calculate_output_dimensions(timepoint, region)
            shapes = omezarr.level_shapes((1, self.num_c, self.num_z, height, width), self.num_pyramid_levels)
        made = []
        row_offset, level_heights = 0, None
        if row_band is not None:       # this call writes one row band of the planes: its own (smaller) level buffers
            level_heights = [s[3] for s in shapes]
            row_offset = int(row_band[0])
            shapes = omezarr.level_shapes((1, self.num_c, self.num_z, int(row_band[1]) - row_offset, shapes[0][4]), len(shapes))
            if len(shapes) != len(level_heights):
                raise ValueError(f"row band {row_band} is too short for {len(level_heights)} pyramid levels")

        def make_writer(batch):
            chunks = self.chunks or (1, 1, 1, 512, 512)
            w = self._stream_writer
            if w is not None and w.matches(shapes, self.dtype, batch, self.zarr_compression, chunks):
                w.retarget(output_path, row_offset, level_heights)      # the same geometry: the next store through the same writer
                made.append(w)
                return w
            self._close_stream_writer()
            key = ('writer', tuple(tuple(s[3:]) for s in shapes), batch, np.dtype(self.dtype).str)
            cached = self._buffer_cache.get(key)
            arena = None
            if cached is None and not self._arena_unsupported:
                # the writer's two slots of level-0 canvases are what the fusion kernel writes on this path: from an arena too
                # (it lives as long as the slot tensors, i.e. with the cached buffers)
                need = 2 * native.canvas_bytes(batch, shapes[0][3], shapes[0][4], native.torch_dtype_of(self.dtype))
                if need >= self.canvas_arena_min_bytes:
                    arena = self._new_arena(need)
            made.append(omezarr.PlaneStreamWriter(output_path, shapes, self.dtype, chunks=self.chunks or (1, 1, 1, 512, 512),
                                                  batch=batch, compression=self.zarr_compression, device=self.device,
                                                  buffers=cached, row_offset=row_offset,
                                                  level_heights=level_heights, canvas_arena=arena))
            self._keep_buffers(key, made[-1].buffers)
            self._stream_writer = made[-1]
            return made[-1]

        before = self._stream_writer.bytes_written if self._stream_writer is not None else 0
        _, ids = self.stitch_planes(timepoint, region, only_planes, progress_callback, stream_to=make_writer, row_band=row_band)
        # bytes of this region's chunks (under run() the writer is drained at the end: the count then lags by what is in flight)
        self.last_bytes_written = sum(w.bytes_written for w in set(made)) - (before if self._stream_writer in made else 0)
        return output_path

    def close(self) -> None:
        """Give back what the instance holds beyond its Python objects: the stream writer's threads, the canvas arena."""
        self._close_stream_writer()
        if self._arena is not None and not self._arena.in_use():
            self._arena.close()
        self._arena = None

    def __del__(self):
        try:
            self._close_stream_writer()
        except Exception:
            pass

    def _close_stream_writer(self) -> None:
        """Everything submitted is on disk and the writer's threads are gone (end of run(), a change of geometry, an error)."""
        w, self._stream_writer = getattr(self, '_stream_writer', None), None
        if w is not None:
            w.close()

    def _run_region_by_planes(self, timepoint, region, rank, world):
        """One region shared by all ranks.  With at least as many (channel, z) planes as ranks, every
        rank takes one contiguous run of planes (sharding.contiguous_blocks: a channel's z planes stay together);
        with fewer, every plane is cut into row bands of 512 * 2^(levels-1) level-0 rows
        (sharding.row_bands: whole chunk rows at every pyramid level) and the (plane, band) units are dealt instead
        -- a rank then reads only the tiles that reach into its bands.  Chunks of an OME-Zarr store span neither
        planes nor bands, so the ranks write into one store without locking."""
        n_planes = self.num_c * self.num_z
        width, height = self.calculate_output_dimensions(timepoint, region)
        bands = sharding.row_bands(height, self.num_pyramid_levels, (self.chunks or (1, 1, 1, 512, 512))[3])
        units = sharding.plane_band_units(n_planes, bands, rank, world)
        print(f"\nProcessing timepoint {timepoint}, region {region}: (plane, band) units {units} (rank {rank}/{world})")
        if rank == 0:
            self.create_region_store(timepoint, region)
        sharding.barrier()
        self.starting_stitching.emit()
        self.starting_saving.emit(False)
        output_path = self._zarr_path(timepoint, region)
        by_band = {}
        for p, b in units:
            by_band.setdefault(b, []).append(p)
        for b, planes in by_band.items():
            output_path = self.stream_region_to_zarr(timepoint, region, planes, progress_callback=self.update_progress.emit,
                                                     create=False, row_band=None if b < 0 else bands[b])
        sharding.barrier()
        return output_path

    def save_region_aics(self, timepoint, region, stitched_region):
        """OME-TIFF (or, for a '.ome.zarr' format, OME-Zarr) output (stitcher.py:691-769) through the
        package-free writers: same path template, channel names / colours, physical pixel sizes."""
        if self.output_format.endswith('.zarr'):
            return self.save_region_ome_zarr(timepoint, region, stitched_region)
        if hasattr(stitched_region, 'cpu'):
            stitched_region = stitched_region.cpu().numpy()
        output_path = os.path.join(self.output_folder, f"{timepoint}_stitched", f"{region}_stitched{self.output_format}")
        os.makedirs(os.path.dirname(output_path), exist_ok=True)
        print(f"Writing OME-TIFF to: {output_path}")
        dz_um = float(self.acquisition_params.get('dz(um)', 1.0)) if self.acquisition_params else 1.0
        write_ome_tiff(output_path, np.asarray(stitched_region), pixel_size_um=self.pixel_size_um, dz_um=dz_um,
                       channel_names=self.monochrome_channels, channel_colors=self.monochrome_colors,
                       name=f"{region}_t{timepoint}")
        return output_path

    def _write_shift_table(self, n_units, my_rows, rank, world, coll, shared: bool = False) -> None:
        """``shift_table.json`` in the output folder: the shifts every (timepoint, region) was fused with.
        With per-region registration every rank contributes the rows it measured: one all-gather of
        ceil(units / world) rows of 8 int32 per rank (RCCL over xGMI with the nccl backend) -- the only
        collective on the path."""
        units = [(int(t), region) for t in self.timepoints for region in self.regions]
        if self.per_region_registration:
            # rows any rank can hold: its block-cyclic share of the units -- or all of them on rank 0 when the
            # ranks share every region plane by plane (fewer units than GPUs) and rank 0 registers each
            per_rank = n_units if shared else -(-n_units // world)
            local = np.zeros((per_rank, sharding.SHIFT_ROW), dtype=np.int32)
            index = np.full(per_rank, -1, dtype=np.int64)
            for slot, (i, row) in enumerate(sorted(my_rows.items())):
                local[slot], index[slot] = row, i
            # the unit index travels in the row's spare high bits of column 0: valid flags use bits 0-1
            local[:, 0] |= ((index + 1).astype(np.int32) << 8)
            table = sharding.all_gather_shift_table(local, device=coll)
            rows = {}
            for row in table:
                i = (int(row[0]) >> 8) - 1
                if i >= 0:
                    clean = row.copy()
                    clean[0] &= 0xFF
                    rows[i] = sharding.row_to_shifts(clean)
        else:
            rows = {i: self._shifts() for i in range(n_units)}
        if rank != 0:
            return
        entries = []
        for i, (t, region) in enumerate(units):
            s = rows.get(i)
            if s is None:
                continue
            e = {'timepoint': t, 'region': region, 'h_shift': [int(v) for v in s.h_shift], 'v_shift': [int(v) for v in s.v_shift]}
            if s.h_shift_rev is not None:
                e.update(h_shift_rev=[int(v) for v in s.h_shift_rev], h_shift_rev_odd=int(s.h_shift_rev_odd))
            entries.append(e)
        with open(os.path.join(self.output_folder, 'shift_table.json'), 'w') as fh:
            json.dump({'per_region_registration': self.per_region_registration, 'shifts': entries,
                       'flatfield_estimator': (getattr(self, 'flatfield_info', None) or {}).get('estimator')
                       if getattr(self, 'apply_flatfield', False) else None},
                      fh, indent=1)

    # --------------------------------------------------------------------- run
    def run(self):
        """(stitcher.py:1226-1299): metadata, [flatfields], [shifts once], then every
        timepoint x region: stitch + save."""
        stime = time.time()
        self.get_timepoints()
        self.extract_acquisition_parameters()
        self.get_pixel_size()
        self.parse_acquisition_metadata()
        # One process per GPU (torchrun): rank 0 registers, the shift table is all-gathered (RCCL
        # over xGMI with the nccl backend), and the (timepoint, region) units are dealt to the ranks
        # block-cyclically -- they are independent, so no image data is ever exchanged.
        rank, world = sharding.rank_and_world()
        self.output_folder = sharding.broadcast_object(self.output_folder)   # the name embeds datetime.now()
        os.makedirs(self.output_folder, exist_ok=True)
        if self.apply_flatfield:
            if rank == 0:
                print("Calculating flatfields...")
                self.getting_flatfields.emit()
                self.get_flatfields(progress_callback=self.update_progress.emit)
                print("Time to calculate flatfields:", time.time() - stime)
            self.flatfields = sharding.broadcast_object(self.flatfields)   # the estimate samples tiles at random
            self.flatfield_info = sharding.broadcast_object(self.flatfield_info)
            if rank == 0 and self.flatfield_info is not None:
                with open(os.path.join(self.output_folder, 'flatfield_info.json'), 'w') as fh:
                    json.dump(self.flatfield_info, fh, indent=1)
        coll = sharding.collective_device(self)
        # all-pairs registration (--all-pairs-registration) is sharded by PAIR: every rank registers its run of the pair
        # list and the float64 pair table is all-gathered (registration.register_all_pairs_sharded); the reference's
        # centre-pair scheme is three tiles' worth of work and stays on rank 0, its 8-int32 row all-gathered
        pair_sharded = world > 1 and self.all_pairs_registration
        if self.use_registration and not self.per_region_registration:
            if rank == 0 or pair_sharded:
                print(f"\nCalculating shifts on region {self.regions[0]}...")
                self._pair_ranks = (rank, world) if pair_sharded else None
                try:
                    self.calculate_shifts(self.timepoints[0], self.regions[0])
                finally:
                    self._pair_ranks = None
            if world > 1 and not pair_sharded:
                row = sharding.shifts_to_row(self._shifts() if rank == 0 else None)
                table = sharding.all_gather_shift_table(row[None], device=coll)
                self._apply_shifts(sharding.first_valid(table))
        units = [(int(t), region) for t in self.timepoints for region in self.regions]
        n_units = len(units)
        output_path = None
        try:
            self._defer_drain = True      # regions of one geometry stream through ONE writer; it is drained once, below
            output_path = self._run_units(units, n_units, rank, world, coll, pair_sharded)
        finally:
            self._defer_drain = False
            self._close_stream_writer()
        sharding.barrier()
        self.starting_saving.emit(True)
        if self.merge_timepoints or self.merge_hcs_regions:
            print("Note: merging timepoints / HCS regions is an output-format step outside the hot-path scope; "
                  "per-(timepoint, region) stores were written.")
        final_path = os.path.join(self.output_folder, f"{self.timepoints[-1]}_stitched",
                                  f"{self.regions[-1]}_stitched{self.output_format}")
        self.finished_saving.emit(final_path, self.dtype)
        print(f"Total processing time: {time.time() - stime}")

    def _run_units(self, units, n_units, rank, world, coll, pair_sharded):
        """The (timepoint, region) loop of run(): returns the last output path."""
        output_path = None
        shared = world > 1 and n_units < world and self.output_format.endswith('.zarr')
        my_rows = {}      # unit index -> shift row measured by this rank (per-region registration)
        if shared:
            # fewer (timepoint, region) units than GPUs: share each region by (channel, z) plane
            # instead -- every rank fuses its planes and writes their chunks into the common store
            for i, (timepoint, region) in enumerate(units):
                if self.per_region_registration and pair_sharded:
                    # the ranks share this region anyway: they share its pairs too
                    self._pair_ranks = (rank, world)
                    try:
                        self.calculate_shifts(timepoint, region)
                    finally:
                        self._pair_ranks = None
                    if rank == 0:
                        my_rows[i] = sharding.shifts_to_row(self._shifts())
                elif self.per_region_registration:
                    if rank == 0:
                        self.calculate_shifts(timepoint, region)
                        my_rows[i] = sharding.shifts_to_row(self._shifts())
                    row = my_rows.get(i, sharding.shifts_to_row(None))
                    self._apply_shifts(sharding.first_valid(sharding.all_gather_shift_table(row[None], device=coll)))
                output_path = self._run_region_by_planes(timepoint, region, rank, world)
            units = []
        for i in sharding.block_cyclic(len(units), rank, world):
            timepoint, region = units[i]
            rtime = time.time()
            print(f"\nProcessing timepoint {timepoint}, region {region}" + (f" (rank {rank}/{world})" if world > 1 else ""))
            os.makedirs(os.path.join(self.output_folder, f"{timepoint}_stitched"), exist_ok=True)
            if self.per_region_registration:
                self.calculate_shifts(timepoint, region)
                my_rows[i] = sharding.shifts_to_row(self._shifts())
            self.starting_stitching.emit()
            if self.output_format.endswith('.zarr'):
                # fused planes stream to the store batch by batch; saving overlaps stitching
                self.starting_saving.emit(False)
                output_path = self.stream_region_to_zarr(timepoint, region, progress_callback=self.update_progress.emit)
            else:
                stitched_region = self.stitch_region(timepoint, region, progress_callback=self.update_progress.emit)
                self.starting_saving.emit(False)
                output_path = self.save_region_aics(timepoint, region, stitched_region)
            print(f"Completed region {region} (saved to {output_path}): {time.time() - rtime}")
        if self.use_registration:
            self._write_shift_table(n_units, my_rows, rank, world, coll, shared)
        return output_path

"""ctypes binding of libsquidstitch.so (include/squidstitch.h) -- the only way the host
reaches the device kernels.  PyTorch-ROCm tensors are used purely as device-buffer
containers: ``tensor.data_ptr()`` in, nothing torch-typed crosses the C-ABI.

There is no CPU fallback: if the shared object is missing or a call fails, this
module raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C image-stitcher_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SQ_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libsquidstitch.so')

SQ_U8, SQ_U16, SQ_F32, SQ_F64 = 1, 2, 4, 8
SQ_FUSE_OVERWRITE, SQ_FUSE_FEATHER = 0, 1
SQ_NORM_NONE, SQ_NORM_PHASE = 0, 1
SQ_FUSE_FORCE_QUEUES, SQ_FUSE_FORCE_STATIC, SQ_FUSE_NO_PLANE_GROUPS, SQ_FUSE_NO_SEAM_OWNERS, SQ_FUSE_CONSECUTIVE_GROUPS = 1, 2, 4, 8, 16
SQ_VERSION = 108
SQ_ARENA_NATURAL_ORDER = 1
SQ_ARENA_TWO_CLASSES = 2
SQ_ARENA_MAX_CLASSES = 8

RECT_DTYPE = np.dtype([('src_y0', '<i4'), ('src_x0', '<i4'), ('h', '<i4'), ('w', '<i4'),
                       ('dst_y', '<i4'), ('dst_x', '<i4')])
PAIR_DTYPE = np.dtype([('ref_tile', '<i4'), ('mov_tile', '<i4'), ('ref_y0', '<i4'), ('ref_x0', '<i4'),
                       ('mov_y0', '<i4'), ('mov_x0', '<i4')])
RESULT_DTYPE = np.dtype([('coarse', '<i4', (2,)), ('fine', '<i4', (2,)), ('ccmax_re', '<f8'),
                         ('ccmax_im', '<f8'), ('src_amp', '<f8'), ('tgt_amp', '<f8')])
SYNTH_DTYPE = np.dtype([('scene_seed', '<u8'), ('noise_seed', '<u8'), ('oy', '<i8'), ('ox', '<i8')])


class NativeError(RuntimeError):
    """A libsquidstitch call returned a negative status."""


class _FuseArgs(C.Structure):
    _fields_ = [
        ('plan', C.c_void_p), ('table_dev', C.c_void_p), ('table_bytes', C.c_int64),
        ('tile_ptrs_dev', C.c_void_p), ('tile_base_dev', C.c_void_p),
        ('tile_plane_stride', C.c_int64), ('tile_stride', C.c_int64),
        ('n_tiles', C.c_int32), ('tile_h', C.c_int32), ('tile_w', C.c_int32),
        ('tile_pitch', C.c_int32), ('tile_dtype', C.c_int32),
        ('flat_ptrs_dev', C.c_void_p), ('flat_dtype', C.c_int32),
        ('canvas_dev', C.c_void_p), ('canvas_plane_stride', C.c_int64),
        ('canvas_h', C.c_int32), ('canvas_w', C.c_int32), ('canvas_pitch', C.c_int32),
        ('canvas_dtype', C.c_int32), ('n_planes', C.c_int32), ('mode', C.c_int32),
        ('scratch_dev', C.c_void_p), ('scratch_bytes', C.c_int64),
        ('flags', C.c_int32), ('grid_blocks', C.c_int32),
    ]


class _RegisterArgs(C.Structure):
    _fields_ = [
        ('tile_ptrs_dev', C.c_void_p), ('tile_base_dev', C.c_void_p), ('tile_stride', C.c_int64),
        ('n_tiles', C.c_int32), ('tile_h', C.c_int32), ('tile_w', C.c_int32),
        ('tile_pitch', C.c_int32), ('tile_dtype', C.c_int32),
        ('minmax_dev', C.c_void_p), ('pairs_dev', C.c_void_p), ('n_pairs', C.c_int32),
        ('n0', C.c_int32), ('n1', C.c_int32), ('upsample_factor', C.c_int32), ('normalization', C.c_int32),
        ('results_dev', C.c_void_p), ('workspace_dev', C.c_void_p), ('workspace_bytes', C.c_int64),
    ]


class _ArenaInfo(C.Structure):
    _fields_ = [('base_dev', C.c_void_p), ('bytes', C.c_int64), ('slice_bytes', C.c_int64), ('n_slices', C.c_int32),
                ('n_candidates', C.c_int32), ('n_classes', C.c_int32), ('class_slices', C.c_int32 * 8),
                ('class_candidates', C.c_int32 * 8), ('interleaved', C.c_int32),
                ('probe_ms', C.c_float), ('create_ms', C.c_float), ('min_pair_gbs', C.c_float), ('max_pair_gbs', C.c_float)]


class _BasicInfo(C.Structure):
    _fields_ = [('reweight_iterations', C.c_int32), ('ladmap_iterations', C.c_int32), ('working_size', C.c_int32)]


EXPORTS = {
    'sq_version': (C.c_int, []),
    'sq_last_error': (C.c_char_p, []),
    'sq_fuse_plan_create': (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'sq_fuse_plan_create_spans': (C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'sq_fuse_plan_expand_scratch_bytes': (C.c_int64, [C.c_void_p]),
    'sq_fuse_plan_expand': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    'sq_fuse_plan_destroy': (None, [C.c_void_p]),
    'sq_fuse_plan_table_bytes': (C.c_int64, [C.c_void_p]),
    'sq_fuse_plan_upload': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'sq_fuse_plan_export': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'sq_fuse_plan_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'sq_fuse_planes': (C.c_int, [C.POINTER(_FuseArgs), C.c_void_p]),
    'sq_tile_minmax': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_void_p, C.c_void_p]),
    'sq_normalize_tiles': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sq_downsample2': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                 C.c_int32, C.c_int32, C.c_void_p]),
    'sq_register_line_supported': (C.c_int, [C.c_int32]),
    'sq_register_workspace_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'sq_register_pairs': (C.c_int, [C.POINTER(_RegisterArgs), C.c_void_p]),
    'sq_selftest_flat_divide': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'sq_selftest_flat_divide_f64': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]),
    'sq_selftest_normalise_divide': (C.c_int, [C.c_void_p, C.c_void_p]),
    'sq_selftest_blend_divide': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'sq_fuse_scratch_bytes': (C.c_int64, [C.c_int32]),
    'sq_arena_create': (C.c_void_p, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(_ArenaInfo)]),
    'sq_arena_info_get': (C.c_int, [C.c_void_p, C.POINTER(_ArenaInfo)]),
    'sq_arena_destroy': (C.c_int, [C.c_void_p]),
    'sq_write_files': (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    'sq_blosc_chunk_count': (C.c_int64, [C.c_int32] * 5),
    'sq_blosc_out_bound': (C.c_int64, [C.c_int32] * 6),
    'sq_blosc_scratch_bytes': (C.c_int64, [C.c_int32] * 6),
    'sq_blosc_encode_planes': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'sq_basic_workspace_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'sq_basic_fit': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(_BasicInfo), C.c_void_p]),
    'sq_synth_tiles': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libsquidstitch.so once; fail loudly when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64.so.7; loading it first makes this library bind to
        # the SAME HIP runtime (same SONAME) instead of bringing /opt/rocm's copy in beside it --
        # two runtimes in one process cannot see each other's device allocations.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(handle, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.sq_version() != SQ_VERSION:
            raise NativeError(f"{LIB_PATH} is version {handle.sq_version()}, this binding expects {SQ_VERSION}: rebuild it")
        _lib = handle
    return _lib


def _check(status: int, what: str) -> None:
    if status < 0:
        raise NativeError(f"{what} failed ({status}): {lib().sq_last_error().decode()}")


def sq_dtype_of(np_dtype) -> int:
    dt = np.dtype(np_dtype)
    table = {np.dtype('uint8'): SQ_U8, np.dtype('uint16'): SQ_U16, np.dtype('float32'): SQ_F32,
             np.dtype('float64'): SQ_F64}
    if dt not in table:
        raise ValueError(f"unsupported dtype {dt} (uint8/uint16 tiles, float32/float64 gains)")
    return table[dt]


def torch_dtype_of(np_dtype):
    import torch
    return {np.dtype('uint8'): torch.uint8, np.dtype('uint16'): torch.uint16,
            np.dtype('float32'): torch.float32, np.dtype('float64'): torch.float64}[np.dtype(np_dtype)]


def np_dtype_of_torch(t) -> np.dtype:
    import torch
    return {torch.uint8: np.dtype('uint8'), torch.uint16: np.dtype('uint16'),
            torch.float32: np.dtype('float32'), torch.float64: np.dtype('float64')}[t]


def _stream_ptr(stream=None) -> int:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


def as_rects(rects) -> np.ndarray:
    """[n, 6] ints (src_y0, src_x0, h, w, dst_y, dst_x) or a RECT_DTYPE array -> RECT_DTYPE."""
    if isinstance(rects, np.ndarray) and rects.dtype == RECT_DTYPE:
        return np.ascontiguousarray(rects)
    a = np.asarray(rects, dtype=np.int64).reshape(-1, 6)
    if a.size and (np.abs(a) >= 2 ** 31).any():
        raise ValueError("rectangle coordinate does not fit int32")
    out = np.zeros(len(a), dtype=RECT_DTYPE)
    for i, name in enumerate(RECT_DTYPE.names):
        out[name] = a[:, i]
    return out


# Largest tile height whose overwrite plan the device can expand (csrc/plan_expand.hip: MAX_BUCKETS row blocks of 8 rows)
EXPAND_MAX_TILE_H = 8176


class FusePlan:
    """Host handle of one fusion plan (sq_fuse_plan_*).  ``table`` is the byte image the
    device reads; ``device_table(device)`` uploads it once and caches the tensor.

    ``expand_on_device=True`` (overwrite plans): the host stops after the sweep into spans and ``device_table`` has the
    work list -- items, seam owners, their order -- produced by kernels in device memory (sq_fuse_plan_create_spans /
    sq_fuse_plan_expand; the table is the host planner's byte for byte, tests/test_plan_gpu.py): what a job pays between
    registration and its first fusion launch drops from ~5.5 ms + a 14.7 MB upload to ~1.5 ms for a 32 x 32 grid.
    ``device_table`` is serialised per plan (the expansion writes the plan's host header), and a table whose expansion or
    upload failed is not kept: the next call starts over."""

    def __init__(self, rects, tile_h: int, tile_w: int, canvas_h: int, canvas_w: int, mode: int = SQ_FUSE_OVERWRITE,
                 expand_on_device: bool = False):
        L = lib()
        self.rects = as_rects(rects)
        self.tile_h, self.tile_w = int(tile_h), int(tile_w)
        self.canvas_h, self.canvas_w = int(canvas_h), int(canvas_w)
        self.mode = int(mode)
        self.n_tiles = len(self.rects)
        self.expand_on_device = bool(expand_on_device) and self.mode == SQ_FUSE_OVERWRITE and self.tile_h <= EXPAND_MAX_TILE_H
        create = L.sq_fuse_plan_create_spans if self.expand_on_device else L.sq_fuse_plan_create
        self._h = create(self.rects.ctypes.data if self.n_tiles else None, self.n_tiles,
                         self.tile_h, self.tile_w, self.canvas_h, self.canvas_w, self.mode)
        if not self._h:
            raise NativeError(f"sq_fuse_plan_create failed: {L.sq_last_error().decode()}")
        self.table_bytes = int(L.sq_fuse_plan_table_bytes(self._h))
        self._table = None
        ns, ni, cv, mr = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        _check(L.sq_fuse_plan_stats(self._h, C.byref(ns), C.byref(ni), C.byref(cv), C.byref(mr)), 'sq_fuse_plan_stats')
        self.n_spans, self.n_items, self.covered_voxels, self.max_refs = ns.value, ni.value, cv.value, mr.value
        self._dev = {}
        import threading
        self._lock = threading.Lock()

    @property
    def handle(self) -> int:
        return self._h

    @property
    def table(self) -> np.ndarray:
        """Host copy of the byte image the device reads (tests, inspection)."""
        if self._table is None:
            if self.expand_on_device:      # the items exist on the device only: read the expanded table back
                if not self._dev:
                    raise NativeError("FusePlan.table: a plan expanded on the device has no host copy before device_table() has run")
                self._table = next(iter(self._dev.values())).cpu().numpy()
            else:
                self._table = np.empty(self.table_bytes, dtype=np.uint8)
                _check(lib().sq_fuse_plan_export(self._h, self._table.ctypes.data, self.table_bytes), 'sq_fuse_plan_export')
        return self._table

    def device_table(self, device):
        """The table in device memory, uploaded once per device (one DMA from the plan's page-locked
        storage; sq_fuse_plan_upload waits for it, so the plan can be dropped at any time)."""
        import torch
        key = str(device)
        with self._lock:      # sq_fuse_plan_expand writes the plan's host header: one expansion (or upload) at a time per plan
            return self._device_table_locked(key, device)

    def _device_table_locked(self, key, device):
        import torch
        if key not in self._dev:
            # on the read-back / upload stream, not the caller's: the copy must not queue behind a fusion
            # launch that is still running there (the entry point waits for the copy, so the table is
            # complete before anything later is launched).  The block comes from THAT stream's pool (see
            # _side_empty): a block of the caller's pool could still be read by a running fusion kernel.
            dev = _side_empty((max(self.table_bytes, 1),), torch.uint8, torch.device(device))
            with torch.cuda.device(dev.device):
                side = _copy_stream(dev.device)
                if self.expand_on_device:
                    need = int(lib().sq_fuse_plan_expand_scratch_bytes(self._h))
                    if need < 0:
                        raise NativeError(f"sq_fuse_plan_expand_scratch_bytes failed: {lib().sq_last_error().decode()}")
                    with torch.cuda.stream(side):      # written and read on the side stream only, dropped on return
                        scratch = torch.empty(max(need, 1), dtype=torch.uint8, device=dev.device)
                    _check(lib().sq_fuse_plan_expand(self._h, dev.data_ptr(), dev.numel(), scratch.data_ptr(), scratch.numel(),
                                                     _stream_ptr(side)), 'sq_fuse_plan_expand')
                    del scratch
                else:
                    _check(lib().sq_fuse_plan_upload(self._h, dev.data_ptr(), dev.numel(), _stream_ptr(side)), 'sq_fuse_plan_upload')
            self._dev[key] = dev
        else:
            self._dev[key].record_stream(torch.cuda.current_stream(self._dev[key].device))
        return self._dev[key]

    def close(self) -> None:
        if getattr(self, '_h', None):
            lib().sq_fuse_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _side_empty(shape, dtype, device):
    """Device tensor that is WRITTEN on the copy stream and READ by kernels on the caller's stream.

    The block is taken from the copy stream's pool and ``record_stream``-ed for the caller's stream, so the
    caching allocator hands it out again only after the caller-stream work enqueued up to the moment it is
    dropped has finished -- and then only to another copy-stream allocation.  (Allocated from the caller's
    pool instead, a dropped block would be reused by the next upload at once and overwritten from the copy
    stream while a fusion kernel launched earlier is still reading it.)"""
    import torch
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    with torch.cuda.stream(_copy_stream(device)):
        t = torch.empty(shape, dtype=dtype, device=device)
    t.record_stream(torch.cuda.current_stream(device))
    return t


def upload_small(host, device):
    """A small host tensor -> device, copied on the read-back / upload stream instead of the caller's: a
    pageable host-to-device copy blocks the host until everything queued before it on ITS stream is done,
    and on the caller's stream that can be a 35 ms fusion launch.  Returns when the data is on the device."""
    import torch
    dst = _side_empty(tuple(host.shape), host.dtype, device)
    if dst.numel():
        with torch.cuda.stream(_copy_stream(dst.device)):
            dst.copy_(host)      # pageable source: returns when the bytes are on the device
    return dst


def pointer_table(tensors: Sequence, device):
    """Device int64 tensor of data_ptr()s (0 for None); the caller keeps ``tensors`` alive."""
    import torch
    ptrs = [0 if t is None else int(t.data_ptr()) for t in tensors]
    return upload_small(torch.tensor(ptrs, dtype=torch.int64), device)


def fuse_planes(plan: FusePlan, tiles, canvas, flats=None, tile_ptrs=None, stream=None, flat_ptrs=None,
                flags: int = 0, grid_blocks: int = 0) -> None:
    """Fuse all planes of ``canvas`` ([P, Hc, Wc] or [..., Hc, Wc] contiguous) from ``tiles``.

    tiles:     contiguous device tensor [P, N, H, W] (N = plan.n_tiles), or None with
               ``tile_ptrs`` = device int64 tensor [P*N] of tile pointers (dense H x W tiles).
    flats:     None, or a list of P device tensors / None (H x W float32 or float64 gains);
               all non-None entries must share one dtype.
    flat_ptrs: optionally ``pointer_table(flats, device)`` made earlier, for callers that fuse with the same
               gains again and again (saves a small blocking upload per call).
    """
    import torch
    L = lib()
    hc, wc = int(canvas.shape[-2]), int(canvas.shape[-1])
    n_planes = int(canvas.numel() // (hc * wc)) if hc * wc else 0
    # contiguous, or [P, Hc, Wc] with dense rows and any row pitch / plane stride (empty_canvas pads the plane stride)
    pitch, plane_stride = wc, hc * wc
    if not canvas.is_cuda:
        raise ValueError("canvas must be a device tensor")
    if not canvas.is_contiguous():
        if canvas.dim() != 3 or (wc > 1 and canvas.stride(2) != 1) or canvas.stride(1) < wc or \
                (n_planes > 1 and canvas.stride(0) < (hc - 1) * canvas.stride(1) + wc):
            raise ValueError("canvas must be contiguous or a [P, Hc, Wc] tensor with unit-stride rows and non-overlapping planes")
        pitch, plane_stride = int(canvas.stride(1)), int(canvas.stride(0))
    a = _FuseArgs()
    a.plan = plan.handle
    table = plan.device_table(canvas.device)
    a.table_dev = table.data_ptr()
    a.table_bytes = table.numel()
    keep = [table]
    if tile_ptrs is not None:
        if tile_ptrs.dtype != torch.int64 or tile_ptrs.numel() != n_planes * plan.n_tiles:
            raise ValueError("tile_ptrs must be int64 with n_planes * n_tiles entries")
        a.tile_ptrs_dev = tile_ptrs.data_ptr()
        a.tile_base_dev = None
        tile_np = np_dtype_of_torch(canvas.dtype) if plan.mode == SQ_FUSE_OVERWRITE else np.dtype('uint16')
        if tiles is not None:
            tile_np = np_dtype_of_torch(tiles.dtype)
    else:
        if tiles is None or not tiles.is_cuda or not tiles.is_contiguous():
            raise ValueError("tiles must be a contiguous device tensor")
        if tiles.numel() != n_planes * plan.n_tiles * plan.tile_h * plan.tile_w:
            raise ValueError(f"tiles has {tiles.numel()} elements, expected "
                             f"{n_planes}x{plan.n_tiles}x{plan.tile_h}x{plan.tile_w}")
        a.tile_ptrs_dev = None
        a.tile_base_dev = tiles.data_ptr()
        a.tile_stride = plan.tile_h * plan.tile_w
        a.tile_plane_stride = plan.n_tiles * plan.tile_h * plan.tile_w
        tile_np = np_dtype_of_torch(tiles.dtype)
    a.n_tiles, a.tile_h, a.tile_w, a.tile_pitch = plan.n_tiles, plan.tile_h, plan.tile_w, plan.tile_w
    a.tile_dtype = sq_dtype_of(tile_np)
    a.flat_ptrs_dev = None
    a.flat_dtype = SQ_F32
    if flats is not None and any(f is not None for f in flats):
        if len(flats) != n_planes:
            raise ValueError("flats needs one entry per plane")
        dts = {f.dtype for f in flats if f is not None}
        if len(dts) != 1:
            raise ValueError("all flatfields must share one dtype")
        for f in flats:
            if f is not None and (tuple(f.shape) != (plan.tile_h, plan.tile_w) or not f.is_contiguous()):
                raise ValueError("flatfield must be a contiguous tile_h x tile_w tensor")
        a.flat_dtype = sq_dtype_of(np_dtype_of_torch(dts.pop()))
        fp = flat_ptrs if flat_ptrs is not None else pointer_table(flats, canvas.device)
        if fp.dtype != torch.int64 or fp.numel() != n_planes:
            raise ValueError("flat_ptrs must be int64 with one entry per plane")
        keep.append(fp)
        a.flat_ptrs_dev = fp.data_ptr()
    if n_planes > 0:   # gain classes + work-queue counters (torch allocations are 512-byte aligned)
        scratch = torch.empty(int(L.sq_fuse_scratch_bytes(n_planes)), dtype=torch.uint8, device=canvas.device)
        keep.append(scratch)
        a.scratch_dev = scratch.data_ptr()
        a.scratch_bytes = scratch.numel()
    a.canvas_dev = canvas.data_ptr()
    a.canvas_plane_stride = plane_stride
    a.canvas_h, a.canvas_w, a.canvas_pitch = hc, wc, pitch
    a.canvas_dtype = sq_dtype_of(np_dtype_of_torch(canvas.dtype))
    a.n_planes = n_planes
    a.mode = plan.mode
    a.flags, a.grid_blocks = int(flags), int(grid_blocks)
    if stream is not None:      # launched on another stream than the one the helpers above allocated for
        for t in keep:
            t.record_stream(stream)
    _check(L.sq_fuse_planes(C.byref(a), _stream_ptr(stream)), 'sq_fuse_planes')


PLANE_ALIGN_BYTES = 128


class _RawDeviceMemory:
    """A range of an arena as an object PyTorch can alias (``__cuda_array_interface__``, uint8); keeps the arena alive."""

    def __init__(self, arena, ptr: int, nbytes: int):
        self._arena = arena
        self.__cuda_array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1', 'data': (int(ptr), False), 'version': 2}


class DeviceArena:
    """Device memory whose every stretch lies over all memory classes of the card (sq_arena_create, csrc/arena.hip): the
    home of fusion canvases.  An MI355X's memory falls into a few classes of tens of GiB each (thirds of the card where it was scanned); the fusion kernel's
    row-segment writes run at 0.55 of the HBM peak inside one class and at 0.73-0.76 spread over them, and a plain
    allocation gets runs of tens of GiB of ONE class.  The arena takes physical slices, measures their classes and maps them
    round-robin into one virtual range, so a canvas carved from it is fast wherever it starts.

    ``take(nbytes)`` bump-allocates (512-byte aligned) and returns a uint8 tensor aliasing the range; ``reset()`` starts
    over (the caller makes sure nothing uses the old tensors any more); ``close()`` unmaps and releases everything.
    ``natural_order`` (no probe, slices as they come) and ``two_classes`` (the two largest classes only) are the controls of
    A/B measurements."""

    def __init__(self, nbytes: int, device=None, slice_bytes: int = 0, unit_bytes: int = 0, natural_order: bool = False, stream=None,
                 candidate_bytes: Optional[int] = None, two_classes: bool = False):
        import torch
        L = lib()
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        info = _ArenaInfo()
        with torch.cuda.device(self.device):
            if candidate_bytes is None:
                # memory comes in runs of tens of GiB of one class: the call may take what is free (but for a reserve), chunk by
                # chunk, until three classes hold a third of the arena each (after 2.5 x the arena: two classes half each), and
                # gives the rest back (create the arena FIRST, while the card is empty: it then takes 1.5-2.5 times its size; the
                # driver clears what it hands out and what it takes back -- 0.5-6 s for 80 GiB, once per process)
                free = torch.cuda.mem_get_info(self.device)[0]
                candidate_bytes = max(int(nbytes), min(8 * int(nbytes), free - (6 << 30)))
            self._h = L.sq_arena_create(int(nbytes), int(candidate_bytes), int(slice_bytes), int(unit_bytes),
                                        (SQ_ARENA_NATURAL_ORDER if natural_order else 0) | (SQ_ARENA_TWO_CLASSES if two_classes else 0), _stream_ptr(stream), C.byref(info))
        if not self._h:
            raise NativeError(f"sq_arena_create({nbytes} bytes) failed: {L.sq_last_error().decode()}")
        self.base, self.nbytes = int(info.base_dev), int(info.bytes)
        self.info = {'bytes': self.nbytes, 'slice_bytes': int(info.slice_bytes), 'n_slices': int(info.n_slices),
                     'n_candidates': int(info.n_candidates), 'n_classes': int(info.n_classes),
                     'class_slices': [int(v) for v in info.class_slices[:max(1, info.n_classes)]],
                     'class_candidates': [int(v) for v in info.class_candidates[:max(1, info.n_classes)]],
                     'interleaved': bool(info.interleaved), 'probe_ms': round(float(info.probe_ms), 2),
                     'create_ms': round(float(info.create_ms), 1), 'min_pair_gbs': round(float(info.min_pair_gbs), 1),
                     'max_pair_gbs': round(float(info.max_pair_gbs), 1)}
        self._used = 0
        self._holders = []

    def take(self, nbytes: int, align: int = 512):
        """uint8 device tensor of ``nbytes`` bytes at the next ``align``-byte boundary of the arena."""
        import torch
        if not self._h:
            raise NativeError("DeviceArena is closed")
        start = -(-self._used // align) * align
        if start + nbytes > self.nbytes:
            raise NativeError(f"DeviceArena: {nbytes} bytes asked, {self.nbytes - start} of {self.nbytes} left")
        self._used = start + int(nbytes)
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        # PyTorch keeps the holder alive for as long as the tensor's STORAGE lives (views included) and nobody else holds it:
        # a weak reference to it says whether the range is still in use
        import weakref
        holder = _RawDeviceMemory(self, self.base + start, nbytes)
        self._holders.append(weakref.ref(holder))
        return torch.as_tensor(holder, device=self.device)

    def in_use(self) -> bool:
        """True while any tensor taken since the last ``reset()`` (or a view of one) is alive."""
        self._holders = [h for h in self._holders if h() is not None]
        return bool(self._holders)

    @property
    def free_bytes(self) -> int:
        return self.nbytes - self._used

    def reset(self) -> None:
        self._used = 0
        self._holders = []

    def close(self) -> None:
        if getattr(self, '_h', None):
            lib().sq_arena_destroy(self._h)      # waits for the device: nothing may be writing into memory that goes away
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def empty_canvas(n_planes: int, hc: int, wc: int, dtype, device, arena: Optional['DeviceArena'] = None):
    """Uninitialised device canvas [n_planes, hc, wc] with dense rows (pitch = wc, like the reference's array)
    whose PLANE stride is rounded up to a multiple of 128 bytes: every plane then starts on a cache-line
    boundary, so the rows of all planes sit at the same phase inside a line and the fusion kernel can carry the
    planes that share a gain image through an item together (fuse.hip, plane groups).  Each plane ``canvas[p]`` is
    contiguous; the tensor as a whole is not.  ``arena``: carve it from this DeviceArena (memory mapped over all
    memory classes of the card: where the fusion kernel writes fastest) instead of a plain allocation."""
    import torch
    esz = torch.empty((), dtype=dtype).element_size()
    unit = PLANE_ALIGN_BYTES // esz
    stride = -(-(hc * wc) // unit) * unit
    if arena is not None:
        flat = arena.take(max(1, n_planes * stride) * esz).view(dtype)
    else:
        flat = torch.empty(max(1, n_planes * stride), dtype=dtype, device=device)
    return flat.as_strided((n_planes, hc, wc), (stride, wc, 1))


def canvas_bytes(n_planes: int, hc: int, wc: int, dtype) -> int:
    """Bytes ``empty_canvas`` takes for this shape (plane stride padded to 128 bytes, start aligned to 512)."""
    import torch
    esz = torch.empty((), dtype=dtype).element_size()
    unit = PLANE_ALIGN_BYTES // esz
    return max(1, n_planes * (-(-(hc * wc) // unit) * unit)) * esz + 512


def planes_to_host(canvas):
    """Device planes [P, Hc, Wc] (any plane stride) -> contiguous numpy array, one D2H copy per plane (a strided
    ``.cpu()`` would first build a contiguous copy ON the device)."""
    import torch
    out = torch.empty(tuple(canvas.shape), dtype=canvas.dtype)
    for p in range(canvas.shape[0]):
        out[p].copy_(canvas[p])
    return out.numpy()


def _tile_table(tiles, tile_ptrs, shape, np_dtype):
    """(ptrs_dev, base_dev, stride, n, h, w, sq dtype) for a stack [N, H, W] or a pointer table."""
    if tile_ptrs is not None:
        n = int(tile_ptrs.numel())
        h, w = (int(v) for v in shape)
        return tile_ptrs.data_ptr(), None, 0, n, h, w, sq_dtype_of(np_dtype)
    n, h, w = (int(v) for v in tiles.shape)
    return None, tiles.data_ptr(), h * w, n, h, w, sq_dtype_of(np_dtype_of_torch(tiles.dtype))


def tile_minmax(tiles, stream=None, tile_ptrs=None, shape=None, np_dtype=None):
    """Per-tile (min, max) of a contiguous device stack [N, H, W] (or of the dense H x W tiles a
    device int64 pointer table names) -> device int32 tensor [N, 2] (uint32 on the C side; the
    values fit int32 for uint8/uint16 tiles)."""
    import torch
    L = lib()
    ptrs, base, stride, n, h, w, dt = _tile_table(tiles, tile_ptrs, shape, np_dtype)
    device = tile_ptrs.device if tile_ptrs is not None else tiles.device
    out = torch.empty((n, 2), dtype=torch.int32, device=device)
    _check(L.sq_tile_minmax(ptrs, base, stride, n, h, w, w, dt, out.data_ptr(), _stream_ptr(stream)),
           'sq_tile_minmax')
    return out


def normalize_tiles(tiles, minmax=None, stream=None):
    """normalize_image on a contiguous device stack [N, H, W] -> normalised stack of the same dtype."""
    import torch
    L = lib()
    n, h, w = (int(v) for v in tiles.shape)
    if minmax is None:
        minmax = tile_minmax(tiles, stream)
    out = torch.empty_like(tiles)
    _check(L.sq_normalize_tiles(None, tiles.data_ptr(), h * w, n, h, w, w, sq_dtype_of(np_dtype_of_torch(tiles.dtype)),
                                minmax.data_ptr(), out.data_ptr(), _stream_ptr(stream)), 'sq_normalize_tiles')
    return out


def downsample2(planes, out=None, stream=None):
    """Next pyramid level of ``planes`` [n, h, w] (uint8 / uint16 device tensor, any row pitch) ->
    [n, h // 2, w // 2]: out[p, y, x] = planes[p, 2y+1, 2x+1], what ome_zarr's Scaler.nearest computes per
    level (stitcher.py:797-798).  ``out`` may be a preallocated tensor (any row pitch)."""
    import torch
    L = lib()
    if planes.dim() != 3 or planes.device.type != 'cuda':
        raise ValueError("planes must be a [n, h, w] device tensor")
    if planes.stride(2) != 1 and planes.shape[2] > 1:
        raise ValueError("planes rows must be contiguous")
    n, h, w = (int(v) for v in planes.shape)
    if out is None:
        out = torch.empty((n, h // 2, w // 2), dtype=planes.dtype, device=planes.device)
    if tuple(out.shape) != (n, h // 2, w // 2) or out.dtype != planes.dtype or out.device != planes.device:
        raise ValueError(f"out must be a {(n, h // 2, w // 2)} tensor of the input's dtype and device")
    if out.numel() and out.stride(2) != 1 and out.shape[2] > 1:
        raise ValueError("out rows must be contiguous")
    if out.numel() == 0:
        return out
    _check(L.sq_downsample2(planes.data_ptr(), planes.stride(0), h, w, planes.stride(1), out.data_ptr(), out.stride(0),
                            out.stride(1), n, sq_dtype_of(np_dtype_of_torch(planes.dtype)), _stream_ptr(stream)),
           'sq_downsample2')
    return out


_COPY_STREAMS = {}


def _copy_stream(device):
    """One stream per device for small result read-backs, so that they do not queue behind whatever the
    caller has enqueued on its own stream since (e.g. a 35 ms fusion launch)."""
    import torch
    key = (device.type, device.index)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COPY_STREAMS[key]


class PendingRegistration:
    """Results of an enqueued sq_register_pairs; ``fetch()`` waits for THAT batch (not for later work on
    the caller's stream) and returns them."""

    def __init__(self, res_dev, keep, n, done=None):
        self._res, self._keep, self._n, self._done = res_dev, keep, n, done

    def fetch(self) -> np.ndarray:
        import torch
        if self._n == 0:
            return np.zeros(0, dtype=RESULT_DTYPE)
        if self._done is not None:
            side = _copy_stream(self._res.device)
            side.wait_event(self._done)
            self._res.record_stream(side)
            with torch.cuda.stream(side):
                host = self._res.cpu()
        else:
            host = self._res.cpu()
        out = host.numpy().view(RESULT_DTYPE).copy()
        self._keep = None
        return out


def register_pairs_async(tiles, minmax, pairs: np.ndarray, n0: int, n1: int, upsample_factor: int = 10,
                         normalization: int = SQ_NORM_PHASE, stream=None, tile_ptrs=None, shape=None,
                         np_dtype=None) -> PendingRegistration:
    """Enqueue batched phase cross-correlation of crop pairs and return without synchronising.
    ``tiles`` [N, H, W] device stack (or a pointer table + shape + dtype), ``pairs`` PAIR_DTYPE."""
    import torch
    L = lib()
    ptrs, base, stride, n, h, w, dt = _tile_table(tiles, tile_ptrs, shape, np_dtype)
    device = tile_ptrs.device if tile_ptrs is not None else tiles.device
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    npairs = len(pairs)
    if npairs == 0:
        return PendingRegistration(None, None, 0)
    ws_bytes = L.sq_register_workspace_bytes(npairs, n0, n1, upsample_factor)
    if ws_bytes < 0:
        raise NativeError(f"sq_register_workspace_bytes failed: {L.sq_last_error().decode()}")
    if min(pairs['ref_tile'].min(), pairs['mov_tile'].min()) < 0 or \
            max(pairs['ref_tile'].max(), pairs['mov_tile'].max()) >= n:
        raise ValueError("pair refers to a tile outside the table")
    if ((pairs['ref_y0'] < 0).any() or (pairs['ref_x0'] < 0).any() or (pairs['mov_y0'] < 0).any()
            or (pairs['mov_x0'] < 0).any() or (pairs['ref_y0'] + n0 > h).any()
            or (pairs['mov_y0'] + n0 > h).any() or (pairs['ref_x0'] + n1 > w).any()
            or (pairs['mov_x0'] + n1 > w).any()):
        raise ValueError("crop reaches outside its tile")
    ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=device)
    pairs_dev = upload_small(torch.from_numpy(pairs.view(np.uint8).reshape(-1)), device)
    res_dev = torch.zeros(npairs * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=device)
    a = _RegisterArgs()
    a.tile_ptrs_dev = ptrs
    a.tile_base_dev = base
    a.tile_stride = stride
    a.n_tiles, a.tile_h, a.tile_w, a.tile_pitch = n, h, w, w
    a.tile_dtype = dt
    a.minmax_dev = minmax.data_ptr()
    a.pairs_dev = pairs_dev.data_ptr()
    a.n_pairs = npairs
    a.n0, a.n1 = int(n0), int(n1)
    a.upsample_factor = int(upsample_factor)
    a.normalization = int(normalization)
    a.results_dev = res_dev.data_ptr()
    a.workspace_dev = ws.data_ptr()
    a.workspace_bytes = ws.numel()
    _check(L.sq_register_pairs(C.byref(a), _stream_ptr(stream)), 'sq_register_pairs')
    done = torch.cuda.Event()
    done.record(stream if stream is not None else torch.cuda.current_stream())
    return PendingRegistration(res_dev, (ws, pairs_dev, minmax, tiles, tile_ptrs), npairs, done)


def register_pairs(tiles, minmax, pairs: np.ndarray, n0: int, n1: int, upsample_factor: int = 10,
                   normalization: int = SQ_NORM_PHASE, stream=None, tile_ptrs=None, shape=None,
                   np_dtype=None) -> np.ndarray:
    """register_pairs_async + fetch: returns a RESULT_DTYPE host array (synchronises)."""
    return register_pairs_async(tiles, minmax, pairs, n0, n1, upsample_factor, normalization, stream,
                                tile_ptrs, shape, np_dtype).fetch()


def write_files(paths: Sequence[str], data: np.ndarray, data_offsets: np.ndarray, n_threads: int = 16) -> int:
    """Write ``len(paths)`` files with native threads (sq_write_files): file i holds ``data[data_offsets[i]:data_offsets[i + 1]]``
    (``data`` a contiguous uint8 array, ``data_offsets`` int64 with one more entry than there are paths).  The directories must
    exist.  Returns the bytes written.  What the chunk writer of the OME-Zarr store uses: tens of thousands of half-MB files per
    batch, which Python's own open / write / close serialised under the interpreter lock."""
    n = len(paths)
    if n == 0:
        return 0
    data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    offs = np.ascontiguousarray(data_offsets, dtype=np.int64)
    if offs.shape != (n + 1,) or offs[0] < 0 or offs[-1] > data.size:
        raise ValueError("data_offsets needs len(paths) + 1 non-decreasing entries inside the data")
    encoded = [os.fsencode(p) + b'\0' for p in paths]
    poffs = np.zeros(n, dtype=np.int64)
    np.cumsum([len(e) for e in encoded[:-1]], out=poffs[1:])
    blob = b''.join(encoded)
    done = C.c_int64(0)
    _check(lib().sq_write_files(blob, poffs.ctypes.data, data.ctypes.data, offs.ctypes.data, n, int(n_threads), C.byref(done)),
           'sq_write_files')
    return int(done.value)


class BloscBuffers:
    """Device buffers of one sq_blosc_encode_planes geometry: scratch, chunk offsets, packed frames, status."""

    def __init__(self, n_planes: int, h: int, w: int, np_dtype, chunk_h: int, chunk_w: int, device):
        import torch
        L = lib()
        dt = sq_dtype_of(np_dtype)
        self.geometry = (int(n_planes), int(h), int(w), dt, int(chunk_h), int(chunk_w))
        self.n_chunks = int(L.sq_blosc_chunk_count(n_planes, h, w, chunk_h, chunk_w))
        self.bound = int(L.sq_blosc_out_bound(*self.geometry))
        need = int(L.sq_blosc_scratch_bytes(*self.geometry))
        if min(self.n_chunks, self.bound, need) < 0:
            raise NativeError(f"sq_blosc geometry: {L.sq_last_error().decode()}")
        self.scratch = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
        self.offsets = torch.empty(self.n_chunks + 1, dtype=torch.int64, device=device)
        self.out = torch.empty(max(self.bound, 1), dtype=torch.uint8, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)


def blosc_encode_planes(planes, chunk_h: int, chunk_w: int, buffers: Optional[BloscBuffers] = None, stream=None) -> BloscBuffers:
    """Enqueue the Blosc-1 (shuffle + LZ4) encoding of the chunks of ``planes`` [n, H, W] (uint8 / uint16 device tensor,
    unit-stride rows, any pitch / plane stride).  Returns the buffers: after the stream has run, chunk i (plane-major,
    chunk row, chunk column) is ``buffers.out[offsets[i]:offsets[i + 1]]`` with ``offsets = buffers.offsets``
    (size 0 = all-zero chunk).  Nothing is synchronised or copied here."""
    L = lib()
    if planes.dim() != 3 or not planes.is_cuda or (planes.shape[2] > 1 and planes.stride(2) != 1):
        raise ValueError("planes must be a [n, H, W] device tensor with unit-stride rows")
    n, h, w = (int(v) for v in planes.shape)
    npdt = np_dtype_of_torch(planes.dtype)
    if buffers is None or buffers.geometry != (n, h, w, sq_dtype_of(npdt), int(chunk_h), int(chunk_w)):
        buffers = BloscBuffers(n, h, w, npdt, chunk_h, chunk_w, planes.device)
    _check(L.sq_blosc_encode_planes(planes.data_ptr(), planes.stride(0) if n > 1 else h * planes.stride(1), planes.stride(1), n, h, w,
                                    sq_dtype_of(npdt), int(chunk_h), int(chunk_w), buffers.scratch.data_ptr(),
                                    buffers.scratch.numel(), buffers.offsets.data_ptr(), buffers.out.data_ptr(), buffers.out.numel(),
                                    buffers.status.data_ptr(), _stream_ptr(stream)), 'sq_blosc_encode_planes')
    return buffers


def basic_fit(tiles, smoothness_flatfield: float = 1.0, stream=None):
    """Flatfield estimate of a device stack [n, H, W] (uint8 / uint16, n <= 80) -> (device float32 [H, W], info dict).
    Replaces ``BaSiC(get_darkfield=False, smoothness_flatfield=...).fit(images).flatfield`` (stitcher.py:374-377);
    parity with basicpy is UNPINNED (the package is absent offline) -- see include/squidstitch.h.  Synchronises."""
    import torch
    L = lib()
    if tiles.dim() != 3 or not tiles.is_cuda or not tiles.is_contiguous():
        raise ValueError("tiles must be a contiguous [n, H, W] device tensor")
    n, h, w = (int(v) for v in tiles.shape)
    need = L.sq_basic_workspace_bytes(n, h, w)
    if need < 0:
        raise NativeError(f"sq_basic_workspace_bytes failed: {L.sq_last_error().decode()}")
    ws = torch.empty(int(need), dtype=torch.uint8, device=tiles.device)
    out = torch.empty((h, w), dtype=torch.float32, device=tiles.device)
    info = _BasicInfo()
    _check(L.sq_basic_fit(None, tiles.data_ptr(), h * w, n, h, w, w, sq_dtype_of(np_dtype_of_torch(tiles.dtype)),
                          float(smoothness_flatfield), out.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(info),
                          _stream_ptr(stream)), 'sq_basic_fit')
    return out, {'reweight_iterations': info.reweight_iterations, 'ladmap_iterations': info.ladmap_iterations,
                 'working_size': info.working_size}


def selftest_flat_divide(exponent: int, n_binades: int, negative: bool, device) -> int:
    """Mismatches between the fast and the IEEE flatfield divide over whole binades of gains (tests)."""
    import torch
    out = torch.zeros(1, dtype=torch.int64, device=device)
    _check(lib().sq_selftest_flat_divide(int(exponent), int(n_binades), int(bool(negative)), out.data_ptr(),
                                         _stream_ptr()), 'sq_selftest_flat_divide')
    return int(out.item())


def selftest_flat_divide_f64(exponent: int, n_binades: int, negative: bool, seed: int, device) -> int:
    """Mismatches between the shortened and the IEEE float64 flatfield divide on random gains (tests)."""
    import torch
    out = torch.zeros(1, dtype=torch.int64, device=device)
    _check(lib().sq_selftest_flat_divide_f64(int(exponent), int(n_binades), int(bool(negative)), int(seed) & (2 ** 64 - 1),
                                             out.data_ptr(), _stream_ptr()), 'sq_selftest_flat_divide_f64')
    return int(out.item())


def selftest_normalise_divide(device) -> int:
    """Mismatches between the registration kernels' shortened normalisation quotient and the IEEE float64 division over
    every (numerator, range) pair of 16-bit integers (tests)."""
    import torch
    out = torch.zeros(1, dtype=torch.int64, device=device)
    _check(lib().sq_selftest_normalise_divide(out.data_ptr(), _stream_ptr()), 'sq_selftest_normalise_divide')
    return int(out.item())


def selftest_blend_divide(exponent: int, n_binades: int, negative: bool, device) -> int:
    """Mismatches between the grouped feather blend's final division and the IEEE one over whole binades (tests)."""
    import torch
    out = torch.zeros(1, dtype=torch.int64, device=device)
    _check(lib().sq_selftest_blend_divide(int(exponent), int(n_binades), int(bool(negative)), out.data_ptr(),
                                          _stream_ptr()), 'sq_selftest_blend_divide')
    return int(out.item())


def synth_tiles(desc: np.ndarray, tile_h: int, tile_w: int, noise_amp: int, np_dtype, device, out=None, stream=None):
    """Generate tiles on the device from SYNTH_DTYPE descriptors -> [n, H, W] tensor."""
    import torch
    L = lib()
    desc = np.ascontiguousarray(desc, dtype=SYNTH_DTYPE)
    n = len(desc)
    if out is None:
        out = torch.empty((n, tile_h, tile_w), dtype=torch_dtype_of(np_dtype), device=device)
    d = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(device)
    for i0 in range(0, n, 32768):   # grid.z limit
        i1 = min(n, i0 + 32768)
        _check(L.sq_synth_tiles(d.data_ptr() + i0 * SYNTH_DTYPE.itemsize, i1 - i0, tile_h, tile_w, noise_amp,
                                sq_dtype_of(np_dtype), out[i0:].data_ptr(), _stream_ptr(stream)), 'sq_synth_tiles')
    return out

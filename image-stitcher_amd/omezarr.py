"""Minimal OME-Zarr (NGFF 0.4, zarr v2 directory store) writer without the ``zarr`` package.

Layout follows what the reference writes through ome_zarr.writer.write_multiscale
(stitcher.py:771-859): a group with arrays "0".."n-1" (TCZYX), chunks (1,1,1,512,512),
``multiscales`` axes t/c/z/y/x with units and per-level scale [1,1,dz,px*2^l,px*2^l],
and an ``omero`` channel block.  Chunk codecs: ``blosc`` = the reference's default (zarr.storage.default_compressor:
Blosc-1 frames, byte shuffle + LZ4), encoded ON THE DEVICE (csrc/blosc.hip) so that only compressed bytes cross PCIe;
``zlib`` (host threads, stdlib) and ``none`` (raw chunks) remain.  The ``blosc`` / ``numcodecs`` packages are absent
offline: frames are checked by an independent pure-Python reader in the tests.

Pyramid levels are what ome_zarr's ``Scaler.nearest`` produces (level l+1 = level l sampled at
[2y+1, 2x+1], floor-halved shape); they are computed on the device (``native.downsample2``,
csrc/pyramid.hip) -- this module only lays chunks out on disk.  ``PlaneStreamWriter`` is the
"next" row 8(f)1: planes leave the GPU batch by batch (pyramid -> pinned D2H -> compression in
host threads) while the next batch is being fused, so a region never has to exist in host memory.
"""
from __future__ import annotations

import json
import os
import zlib
from typing import List, Optional, Sequence

import numpy as np


def _write_json(path: str, obj) -> None:
    with open(path, 'w') as fh:
        json.dump(obj, fh, indent=1)


def blosc_decode(frame: bytes) -> bytes:
    """Blosc-1 frame -> bytes (LZ4 or memcpy'd frames, byte shuffle; what csrc/blosc.hip and c-blosc's lz4 path emit)."""
    version, versionlz, flags, typesize = frame[0], frame[1], frame[2], frame[3]
    nbytes, blocksize, cbytes = (int.from_bytes(frame[4 + 4 * k:8 + 4 * k], 'little') for k in range(3))
    if version != 2 or cbytes != len(frame):
        raise ValueError("not a Blosc-1 frame")
    if flags & 0x02:
        return bytes(frame[16:16 + nbytes])
    if (flags >> 5) != 1 or versionlz != 1:
        raise ValueError("only LZ4 frames are read here")
    nblocks = -(-nbytes // blocksize) if nbytes else 0
    out = bytearray()
    for b in range(nblocks):
        n = min(blocksize, nbytes - b * blocksize)
        at = int.from_bytes(frame[16 + 4 * b:20 + 4 * b], 'little')
        nsplits = 1 if (flags & 0x10) or typesize > 16 or n // max(typesize, 1) < 128 or n % blocksize else typesize
        raw = bytearray()
        for _ in range(nsplits):
            cb = int.from_bytes(frame[at:at + 4], 'little')
            at += 4
            want = n // nsplits
            raw += frame[at:at + cb] if cb == want else _lz4_block(frame[at:at + cb], want)
            at += cb
        if (flags & 0x01) and typesize > 1:
            nel = n // typesize
            arr = np.frombuffer(bytes(raw[:nel * typesize]), dtype=np.uint8).reshape(typesize, nel).T
            raw = bytearray(arr.tobytes()) + raw[nel * typesize:]
        out += raw
    return bytes(out)


def _lz4_block(src: bytes, n: int) -> bytearray:
    out = bytearray()
    i, end = 0, len(src)
    while i < end:
        token = src[i]
        i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                v = src[i]
                i += 1
                lit += v
                if v != 255:
                    break
        out += src[i:i + lit]
        i += lit
        if i >= end:
            break
        off = src[i] | (src[i + 1] << 8)
        i += 2
        ml = token & 15
        if ml == 15:
            while True:
                v = src[i]
                i += 1
                ml += v
                if v != 255:
                    break
        ml += 4
        start = len(out) - off
        if off == 0 or start < 0:
            raise ValueError("corrupt LZ4 block")
        if off >= ml:
            out += out[start:start + ml]
        else:
            for k in range(ml):
                out.append(out[start + k])
    if len(out) != n:
        raise ValueError(f"LZ4 block decoded to {len(out)} bytes, expected {n}")
    return out


def read_array(path: str) -> np.ndarray:
    """One array of a store back into memory (used by tests to check the store round-trips)."""
    with open(os.path.join(path, '.zarray')) as fh:
        meta = json.load(fh)
    shape, chunks, dt = meta['shape'], meta['chunks'], np.dtype(meta['dtype'])
    codec = (meta.get('compressor') or {}).get('id')
    decode = {None: bytes, 'zlib': zlib.decompress, 'blosc': blosc_decode}[codec]
    out = np.zeros(shape, dtype=dt)
    for idx in np.ndindex(*[-(-s // c) for s, c in zip(shape, chunks)]):
        p = os.path.join(path, *map(str, idx))
        if not os.path.exists(p):
            continue
        with open(p, 'rb') as fh:
            raw = fh.read()
        block = np.frombuffer(decode(raw), dtype=dt).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out


def level_shapes(shape: Sequence[int], num_levels: int) -> List[tuple]:
    """TCZYX shapes of the pyramid levels: every level halves y and x of the one before, rounding
    down (Scaler.nearest resizes to (Y // 2, X // 2))."""
    out = [tuple(int(v) for v in shape)]
    for _ in range(1, max(1, num_levels)):
        t, c, z, y, x = out[-1]
        if y < 2 or x < 2:
            break
        out.append((t, c, z, y // 2, x // 2))
    return out


def _compressor(compression: str, level: int = 1):
    if compression in (None, 'none', 'raw'):
        return None
    if compression == 'zlib':
        return {'id': 'zlib', 'level': int(level)}
    if compression == 'blosc':      # what zarr.storage.default_compressor serialises to (zarr 2.x)
        return {'id': 'blosc', 'cname': 'lz4', 'clevel': 5, 'shuffle': 1, 'blocksize': 0}
    raise ValueError(f"compression must be 'blosc', 'zlib' or 'none', got {compression!r}")


def _zarray_meta(shape, chunks, dtype, compression='zlib', level=1):
    dtype = np.dtype(dtype)
    chunks = tuple(int(min(c, s)) if s else int(c) for c, s in zip(chunks, shape))
    return chunks, {
        'zarr_format': 2, 'shape': list(shape), 'chunks': list(chunks),
        'dtype': dtype.newbyteorder('<').str if dtype.itemsize > 1 else dtype.str,
        'compressor': _compressor(compression, level), 'fill_value': 0, 'order': 'C', 'filters': None,
        'dimension_separator': '/'}


def create_store(path: str, shape: Sequence[int], dtype, *, pixel_size_um: float, dz_um: float = 1.0,
                 channel_names: Sequence[str] = (), channel_colors: Sequence[int] = (), num_levels: int = 1,
                 chunks=(1, 1, 1, 512, 512), name: str = 'stitched', compression: str = 'zlib') -> List[tuple]:
    """Group + array metadata of a multiscale OME-Zarr image, no chunks yet.  Returns the level
    shapes.  Chunks are then added plane by plane (``write_plane_levels``), by any number of processes."""
    os.makedirs(path, exist_ok=True)
    _write_json(os.path.join(path, '.zgroup'), {'zarr_format': 2})
    shapes = level_shapes(shape, num_levels)
    datasets = []
    for lv, shp in enumerate(shapes):
        os.makedirs(os.path.join(path, str(lv)), exist_ok=True)
        _write_json(os.path.join(path, str(lv), '.zarray'), _zarray_meta(shp, chunks, dtype, compression)[1])
        sc = 2 ** lv
        datasets.append({'path': str(lv), 'coordinateTransformations': [
            {'type': 'scale', 'scale': [1, 1, dz_um, pixel_size_um * sc, pixel_size_um * sc]}]})
    dt = np.dtype(dtype)
    info = np.iinfo(dt) if np.issubdtype(dt, np.integer) else None
    attrs = {
        'multiscales': [{
            'version': '0.4', 'name': name,
            'axes': [{'name': 't', 'type': 'time', 'unit': 'second'}, {'name': 'c', 'type': 'channel'},
                     {'name': 'z', 'type': 'space', 'unit': 'micrometer'},
                     {'name': 'y', 'type': 'space', 'unit': 'micrometer'},
                     {'name': 'x', 'type': 'space', 'unit': 'micrometer'}],
            'datasets': datasets}],
        'omero': {'id': 1, 'name': name, 'version': '0.4', 'channels': [
            {'label': n, 'color': f'{(channel_colors[i] if i < len(channel_colors) else 0xFFFFFF):06X}',
             'window': {'start': 0, 'end': int(info.max) if info else 1, 'min': 0, 'max': int(info.max) if info else 1},
             'active': True, 'coefficient': 1, 'family': 'linear'}
            for i, n in enumerate(channel_names)]},
    }
    _write_json(os.path.join(path, '.zattrs'), attrs)
    return shapes


def chunk_jobs(levels: Sequence[np.ndarray], coords: Sequence[tuple], chunks=(1, 1, 1, 512, 512), row_offset: int = 0,
               level_heights: Optional[Sequence[int]] = None) -> list:
    """One job per chunk of whole (t, c, z) planes: ``levels[l][i]`` is the 2-D plane ``coords[i]`` at
    pyramid level l.  Chunks never span planes (chunk shape (1,1,1,cy,cx)), so different processes can
    write different planes of one store concurrently.

    ``row_offset`` > 0: the arrays are one ROW BAND of the planes, starting at that level-0 row (sharding.row_bands:
    a multiple of chunk rows x 2^(levels-1), so the band starts on a chunk-row boundary of every level);
    ``level_heights`` are then the heights of the FULL levels (the chunk height of a level is min(512, its height))."""
    if tuple(chunks[:3]) != (1, 1, 1):
        raise ValueError("plane-wise writing needs chunks of shape (1, 1, 1, cy, cx)")
    jobs = []
    for lv, arr in enumerate(levels):
        if arr.ndim != 3 or len(arr) < len(coords):
            raise ValueError(f"level {lv}: expected [n_planes, y, x] with n_planes >= {len(coords)}, got {arr.shape}")
        full_h = arr.shape[1] if level_heights is None else int(level_heights[lv])
        cy, cx = min(chunks[3], full_h), min(chunks[4], arr.shape[2])
        if cy == 0 or cx == 0:
            continue
        y_off = int(row_offset) >> lv
        if y_off % cy:
            raise ValueError(f"row band at level-0 row {row_offset} does not start on a chunk row of level {lv}")
        for i, (t, c, z) in enumerate(coords):
            for y in range(0, arr.shape[1], cy):
                for x in range(0, arr.shape[2], cx):
                    jobs.append((lv, t, c, z, (y_off + y) // cy, x // cx, arr[i], y, x, cy, cx))
    return jobs


HOST_CODECS = (None, 'none', 'raw', 'zlib')


def emit_chunk(path: str, job, compression: str = 'zlib', level: int = 1) -> int:
    """Write one chunk (all-zero chunks are left to fill_value).  Returns the bytes written."""
    lv, t, c, z, iy, ix, plane, y, x, cy, cx = job
    if compression not in HOST_CODECS:
        # 'blosc' chunks are encoded on the device (PlaneStreamWriter / write_ome_zarr): writing zlib bytes under a
        # .zarray that says blosc would make a store nothing can read
        raise ValueError(f"emit_chunk compresses on the host with {sorted(str(c) for c in HOST_CODECS)}; {compression!r} chunks "
                         "come from the device encoder (write_ome_zarr / PlaneStreamWriter)")
    block = plane[y:y + cy, x:x + cx]
    if not block.any():
        return 0
    if block.shape != (cy, cx):
        full = np.zeros((cy, cx), dtype=plane.dtype)
        full[:block.shape[0], :block.shape[1]] = block
        block = full
    cdir = os.path.join(path, str(lv), str(t), str(c), str(z), str(iy))
    os.makedirs(cdir, exist_ok=True)
    raw = memoryview(np.ascontiguousarray(block)).cast('B')     # no second copy: file write / zlib read the buffer
    data = raw if compression in (None, 'none', 'raw') else zlib.compress(raw, level)
    with open(os.path.join(cdir, str(ix)), 'wb') as fh:
        fh.write(data)
    return len(data)


def write_plane_levels(path: str, levels: Sequence[np.ndarray], coords: Sequence[tuple], chunks=(1, 1, 1, 512, 512),
                       compression: str = 'zlib', level: int = 1, workers: Optional[int] = None, pool=None,
                       row_offset: int = 0, level_heights: Optional[Sequence[int]] = None) -> int:
    """Write the chunks of whole planes (or of one row band of them), every pyramid level given (see
    ``chunk_jobs``), into a store made by ``create_store``.  Returns the bytes written."""
    from concurrent.futures import ThreadPoolExecutor
    if compression not in HOST_CODECS:
        raise ValueError(f"write_plane_levels compresses on the host with {sorted(str(c) for c in HOST_CODECS)}; {compression!r} "
                         "chunks come from the device encoder (write_ome_zarr / PlaneStreamWriter)")
    jobs = chunk_jobs(levels, coords, chunks, row_offset, level_heights)
    if pool is not None:
        return sum(pool.map(lambda j: emit_chunk(path, j, compression, level), jobs))
    n = workers if workers is not None else min(32, os.cpu_count() or 4)
    if n <= 1 or len(jobs) < 4:
        return sum(emit_chunk(path, j, compression, level) for j in jobs)
    with ThreadPoolExecutor(max_workers=n) as tp:
        return sum(tp.map(lambda j: emit_chunk(path, j, compression, level), jobs))


def device_levels(planes_dev, n_levels: int, out: Optional[list] = None) -> list:
    """[planes_dev] + its n_levels - 1 pyramid levels, computed on the device (sq_downsample2)."""
    from . import native
    levels = [planes_dev]
    for lv in range(1, n_levels):
        levels.append(native.downsample2(levels[-1], None if out is None else out[lv - 1][:len(planes_dev)]))
    return levels


def _pad_planes(full, m: int):
    """A partly filled slot: the encoder's buffers are sized for the whole batch, so the planes past ``m`` are
    encoded too -- zero them first, their chunks then have size 0 and nothing is written for them."""
    full[m:].zero_()
    return full


def write_ome_zarr(path: str, image, *, pixel_size_um: float, dz_um: float = 1.0,
                   channel_names: Sequence[str] = (), channel_colors: Sequence[int] = (),
                   num_levels: int = 1, chunks=(1, 1, 1, 512, 512), name: str = 'stitched', compression: str = 'zlib',
                   device=None) -> str:
    """Write a (T, C, Z, Y, X) array (numpy, or a device tensor) as a multiscale OME-Zarr image.  The
    pyramid levels come from the device kernel, a batch of planes at a time; with ``num_levels`` 1 no
    GPU is touched for a numpy input."""
    if image.ndim != 5:
        raise ValueError(f"expected a 5-D TCZYX array, got {tuple(image.shape)}")
    on_device = hasattr(image, 'data_ptr')
    dtype = np.dtype(str(image.dtype).replace('torch.', '')) if on_device else image.dtype
    shape = tuple(int(v) for v in image.shape)
    shapes = create_store(path, shape, dtype, pixel_size_um=pixel_size_um, dz_um=dz_um, channel_names=channel_names,
                          channel_colors=channel_colors, num_levels=num_levels, chunks=chunks, name=name,
                          compression=compression)
    t_, c_, z_ = shape[:3]
    coords = [(t, c, z) for t in range(t_) for c in range(c_) for z in range(z_)]
    planes = image.reshape((-1,) + shape[3:])
    if len(shapes) == 1 and not on_device and compression != 'blosc':
        write_plane_levels(path, [planes], coords, chunks, compression)
        return path
    import torch
    if not on_device:
        from . import native
        native.lib()   # fail loudly: the pyramid is a device kernel, there is no host decimation
        device = device if device is not None else 'cuda:0'
    batch = max(1, (1 << 30) // max(1, shape[3] * shape[4] * dtype.itemsize))
    with PlaneStreamWriter(path, shapes, dtype, chunks=chunks, batch=batch, compression=compression,
                           device=planes.device if on_device else device) as writer:
        for b0 in range(0, len(coords), batch):
            part = planes[b0:b0 + batch]
            dst = writer.acquire(len(part))
            dst.copy_(part if on_device else torch.from_numpy(np.ascontiguousarray(part)), non_blocking=False)
            writer.submit(coords[b0:b0 + batch])
    return path


class PlaneStreamWriter:
    """Streams fused planes from the device into a store made by ``create_store``.

    Two slots, each = a device canvas of ``batch`` planes, device buffers for its pyramid levels and
    pinned host mirrors of all of them.  ``acquire(m)`` hands out the slot's canvas (waiting until the
    slot's previous contents are on disk); the caller fuses into it on the current stream and calls
    ``submit(coords)``, which enqueues the pyramid kernels behind the fusion and the D2H copies on a
    separate copy stream behind those, and returns at once; a dispatcher thread waits for the copies
    and fans the chunks out to the compression threads.  So batch k is copied out, compressed and
    written while batch k+1 is read, copied in and fused.
    """

    def __init__(self, path: str, shapes: Sequence[tuple], dtype, *, chunks=(1, 1, 1, 512, 512), batch: int = 1,
                 compression: str = 'zlib', level: int = 1, device='cuda:0', workers: Optional[int] = None, slots: int = 2,
                 buffers=None, row_offset: int = 0, level_heights: Optional[Sequence[int]] = None, canvas_arena=None):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        import torch
        from . import native
        self.path, self.chunks, self.compression, self.level = path, tuple(chunks), compression, level
        self.batch = int(batch)
        # a writer of one row band: ``shapes`` are the band's level shapes, chunks land ``row_offset`` level-0 rows down
        self.row_offset, self.level_heights = int(row_offset), (None if level_heights is None else list(level_heights))
        self.bytes_written = 0
        tdtype = native.torch_dtype_of(np.dtype(dtype).type)
        yx = [tuple(s[3:]) for s in shapes]
        # ``buffers``: the (device, pinned host) slot buffers of an earlier writer of the same geometry --
        # page-locking host memory costs more than a small region's whole fusion, so callers that write
        # many regions (one per well and timepoint) hand them on
        self._blosc = compression == 'blosc'
        if buffers is None:
            # level 0 = the fusion canvas: dense rows, planes on 128-byte lines (native.empty_canvas)
            # (canvas_arena: a native.DeviceArena to carve the level-0 canvases from -- memory mapped over all memory classes of
            #  the card, where the fusion kernel writes fastest; the slot tensors keep it alive)
            dev = [[native.empty_canvas(self.batch, s[0], s[1], tdtype, device, arena=canvas_arena) if lv == 0 else
                    torch.empty((self.batch,) + s, dtype=tdtype, device=device) for lv, s in enumerate(yx)] for _ in range(slots)]
            if self._blosc:
                # chunks are encoded on the device (csrc/blosc.hip): the host mirrors hold packed FRAMES, not planes
                enc = [[native.BloscBuffers(self.batch, s[0], s[1], np.dtype(dtype), *self._chunk_yx(lv, s), device)
                        for lv, s in enumerate(yx)] for _ in range(slots)]
                host = [[(torch.empty(e.bound, dtype=torch.uint8, pin_memory=True),
                          torch.empty(e.n_chunks + 1, dtype=torch.int64, pin_memory=True),
                          torch.empty(1, dtype=torch.int32, pin_memory=True)) for e in slot_enc] for slot_enc in enc]
                buffers = (dev, host, enc)
            else:
                buffers = (dev, [[torch.empty((self.batch,) + s, dtype=tdtype, pin_memory=True) for s in yx] for _ in range(slots)])
        self.buffers = buffers
        if self._blosc != (len(buffers) == 3):
            raise ValueError("buffers do not match this writer's compression")
        self._dev, self._host = buffers[0], buffers[1]
        self._enc = buffers[2] if self._blosc else None
        if len(self._dev) != slots or [tuple(t.shape) for t in self._dev[0]] != [(self.batch,) + s for s in yx] \
                or self._dev[0][0].dtype != tdtype:
            raise ValueError("buffers do not match this writer's geometry")
        # D2H copies run on their own stream: the link is full duplex, so batch k leaves the device while
        # batch k+1's tiles arrive and are fused on the caller's stream
        self._copy_stream = torch.cuda.Stream(device=device)
        # ... and the writer thread fetches a slot's packed frames on a stream of ITS own: on _copy_stream they would
        # queue behind the wait for the NEXT slot's fusion (submit() parks that wait there) and the disk write-out the
        # two slots are meant to overlap would serialise
        self._frame_stream = torch.cuda.Stream(device=device)
        self._free = [threading.Event() for _ in range(slots)]
        for e in self._free:
            e.set()
        self._slot = -1
        self._m = 0
        self._error = None
        self._pool = ThreadPoolExecutor(max_workers=workers if workers is not None else min(32, os.cpu_count() or 4))
        self._queue: "queue.Queue" = queue.Queue()
        self._thread = threading.Thread(target=self._dispatch, name='zarr-writer', daemon=True)
        self._thread.start()

    def _dispatch(self):
        while True:
            item = self._queue.get()
            if item is None:
                return
            slot, coords, event, target = item
            path, row_offset, level_heights = target
            try:
                event.synchronize()
                if self._blosc:
                    self.bytes_written += self._write_frames(slot, coords, path, row_offset)
                    continue
                levels = [h.numpy() for h in self._host[slot]]
                self.bytes_written += write_plane_levels(path, levels, coords, self.chunks, self.compression,
                                                         self.level, pool=self._pool, row_offset=row_offset,
                                                         level_heights=level_heights)
            except BaseException as exc:   # surfaced by the next acquire() / close()
                self._error = exc
            finally:
                self._free[slot].set()

    def _chunk_yx(self, lv: int, level_yx) -> tuple:
        """Chunk shape of pyramid level ``lv`` in the STORE: zarr clamps a chunk dimension to the array's -- the full
        level's, also when this writer holds only a row band of it."""
        full_h = level_yx[0] if self.level_heights is None else int(self.level_heights[lv])
        return min(self.chunks[3], full_h), min(self.chunks[4], level_yx[1])

    def _write_frames(self, slot: int, coords: Sequence[tuple], path: str, row_offset: int) -> int:
        """Blosc mode: the chunk offsets of this slot are on the host; fetch exactly the packed frames (compressed bytes
        only cross PCIe) and write one file per non-empty chunk."""
        import torch
        m = len(coords)
        done = torch.cuda.Event()
        totals = []
        with torch.cuda.stream(self._frame_stream):     # the encoder has finished: _dispatch waited for the slot's event
            for enc, (frames, offsets, status) in zip(self._enc[slot], self._host[slot]):
                if int(status[0]):
                    raise RuntimeError("sq_blosc_encode_planes: output buffer too small")
                per_plane = enc.n_chunks // self.batch
                total = int(offsets[m * per_plane])
                totals.append(total)
                if total:
                    frames[:total].copy_(enc.out[:total], non_blocking=True)
            done.record()
        done.synchronize()
        # one native call per pyramid level (native.write_files: open / write / close on native threads, straight from the
        # page-locked frame buffer): the paths and the directories are worked out here, a few hundred directories per batch
        from . import native
        written = 0
        for lv, (enc, (frames, offsets, _)) in enumerate(zip(self._enc[slot], self._host[slot])):
            n_planes_geo, h, w, _, cy, cx = enc.geometry
            ncy, ncx = -(-h // cy), -(-w // cx)
            y_off = row_offset >> lv
            cyf = cy
            if y_off % cyf:
                raise ValueError(f"row band at level-0 row {row_offset} does not start on a chunk row of level {lv}")
            off = offsets.numpy()
            per_plane = ncy * ncx
            sizes = np.diff(off[:len(coords) * per_plane + 1])
            keep = np.nonzero(sizes > 0)[0]
            if not len(keep):
                continue
            paths = []
            made = set()
            for k in keep.tolist():
                i, rem = divmod(k, per_plane)
                iy, ix = divmod(rem, ncx)
                t, c, z = coords[i]
                cdir = os.path.join(path, str(lv), str(t), str(c), str(z), str(y_off // cyf + iy))
                if cdir not in made:
                    os.makedirs(cdir, exist_ok=True)
                    made.add(cdir)
                paths.append(os.path.join(cdir, str(ix)))
            # the kept chunks' byte ranges: frames are packed densely, empty chunks take no bytes, so consecutive kept chunks are
            # contiguous in the buffer and [off[k], off[k + 1]) for the kept k is one non-decreasing offset list
            starts = off[keep]
            ends = off[keep + 1]
            if len(keep) > 1 and not np.array_equal(starts[1:], ends[:-1]):
                raise RuntimeError("blosc frames are not packed densely")
            data_offsets = np.concatenate([starts, ends[-1:]]).astype(np.int64)
            written += native.write_files(paths, frames.numpy(), data_offsets, n_threads=self._pool._max_workers)
        return written

    def _check(self):
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def acquire(self, m: int):
        """Device canvas [m, Hc, Wc] of the next slot (contents undefined)."""
        if not 0 < m <= self.batch:
            raise ValueError(f"a slot holds 1..{self.batch} planes, asked for {m}")
        self._slot = (self._slot + 1) % len(self._dev)
        self._free[self._slot].wait()
        self._check()
        self._m = m
        return self._dev[self._slot][0][:m]

    def submit(self, coords: Sequence[tuple]) -> None:
        """The canvas handed out by the last ``acquire`` is (being) filled on the current stream."""
        import torch
        slot, m = self._slot, self._m
        if len(coords) != m:
            raise ValueError(f"{m} planes acquired, {len(coords)} coordinates given")
        dev = self._dev[slot]
        levels = device_levels(dev[0][:m], len(dev), out=dev[1:])
        if self._blosc:        # encode every level's chunks behind the pyramid, on the caller's stream
            from . import native
            for lv, enc in enumerate(self._enc[slot]):
                full = dev[lv]     # the buffers are sized for a full batch: encode all of it when the batch is full
                native.blosc_encode_planes(full if m == self.batch else _pad_planes(full, m), enc.geometry[4], enc.geometry[5], enc)
        fused = torch.cuda.Event()
        fused.record()                                  # fusion + pyramid (+ encoding) of this slot, on the caller's stream
        event = torch.cuda.Event()
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(fused)
            if self._blosc:
                for enc, (frames, offsets, status) in zip(self._enc[slot], self._host[slot]):
                    offsets.copy_(enc.offsets, non_blocking=True)
                    status.copy_(enc.status, non_blocking=True)
            for d, h in (() if self._blosc else zip(dev, self._host[slot])):
                if d.is_contiguous():
                    h[:m].copy_(d[:m], non_blocking=True)
                else:                       # padded plane stride: every plane is contiguous, the stack is not
                    for p in range(m):
                        h[p].copy_(d[p], non_blocking=True)
            event.record()
        self._free[slot].clear()
        self._queue.put((slot, list(coords), event, (self.path, self.row_offset, self.level_heights)))

    def retarget(self, path: str, row_offset: int = 0, level_heights: Optional[Sequence[int]] = None) -> None:
        """The planes submitted FROM NOW ON belong to another store of the same geometry (the next well, the next timepoint): what
        is still in flight keeps the target it was submitted with.  One writer then serves a whole run, and the chunks of region k
        reach the disk while region k + 1 is read, registered and fused (``drain`` before anything reads the stores)."""
        self.path = path
        self.row_offset, self.level_heights = int(row_offset), (None if level_heights is None else list(level_heights))

    def matches(self, shapes: Sequence[tuple], dtype, batch: int, compression: str, chunks) -> bool:
        """Can this writer take planes of these level shapes (its buffers are sized for one geometry)?"""
        from . import native
        yx = [tuple(s[3:]) for s in shapes]
        return (self._thread.is_alive() and self.batch == int(batch) and self.compression == compression and self.chunks == tuple(chunks)
                and [tuple(t.shape[1:]) for t in self._dev[0]] == yx
                and self._dev[0][0].dtype == native.torch_dtype_of(np.dtype(dtype).type))

    def drain(self) -> None:
        """Wait until everything submitted so far is on disk (the writer stays usable)."""
        for e in self._free:
            e.wait()
        self._check()

    def close(self):
        for e in self._free:
            e.wait()
        if self._thread.is_alive():
            self._queue.put(None)
            self._thread.join()
        self._pool.shutdown(wait=True)
        self._check()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

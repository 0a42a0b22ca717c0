"""Minimal baseline-TIFF reader/writer (numpy only) for Squid tile files.

The reference reads tiles through ``dask_image.imread`` (stitcher.py:19,226,536,654),
i.e. whatever pims/tifffile decode.  Squid writes uncompressed little-endian
grayscale TIFFs (uint8/uint16); that is all this module handles natively.  Any
other file type (bmp/png/jpg, compressed TIFF) is delegated to PIL when present.
No GPU code here: this is host-side ingest either side of the hot path.
"""
from __future__ import annotations

import struct

import numpy as np

_TYPES = {1: 'B', 2: 'c', 3: 'H', 4: 'I', 5: 'II', 16: 'Q'}
_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 16: 8}


def write_tiff(path: str, img: np.ndarray) -> None:
    """Write a 2-D uint8/uint16 (or HxWx3 uint8) array as one uncompressed strip."""
    img = np.ascontiguousarray(img)
    if img.dtype not in (np.uint8, np.uint16):
        raise ValueError(f"unsupported dtype {img.dtype}")
    if img.ndim == 2:
        h, w = img.shape
        spp, photometric = 1, 1
    elif img.ndim == 3 and img.shape[2] == 3:
        h, w, spp = img.shape
        photometric = 2
    else:
        raise ValueError(f"unsupported shape {img.shape}")
    bits = img.dtype.itemsize * 8
    data = img.astype(img.dtype.newbyteorder('<'), copy=False).tobytes()
    entries = [
        (256, 4, 1, w), (257, 4, 1, h),
        (258, 3, spp, None),            # BitsPerSample (patched below)
        (259, 3, 1, 1),                 # no compression
        (262, 3, 1, photometric),
        (273, 4, 1, None),              # StripOffsets
        (277, 3, 1, spp),
        (278, 4, 1, h),                 # RowsPerStrip
        (279, 4, 1, len(data)),         # StripByteCounts
    ]
    n = len(entries)
    ifd_off = 8
    extra_off = ifd_off + 2 + n * 12 + 4
    extra = b''
    if spp == 3:
        bits_off = extra_off
        extra = struct.pack('<3H', bits, bits, bits) + b'\0\0'
    data_off = extra_off + len(extra)
    out = [b'II', struct.pack('<HI', 42, ifd_off), struct.pack('<H', n)]
    for tag, typ, cnt, val in entries:
        if tag == 258:
            val = bits if spp == 1 else bits_off
            if spp == 1:
                out.append(struct.pack('<HHIHH', tag, typ, cnt, val, 0))
            else:
                out.append(struct.pack('<HHII', tag, typ, cnt, val))
            continue
        if tag == 273:
            val = data_off
        if typ == 3:
            out.append(struct.pack('<HHIHH', tag, typ, cnt, val, 0))
        else:
            out.append(struct.pack('<HHII', tag, typ, cnt, val))
    out.append(struct.pack('<I', 0))
    out.append(extra)
    out.append(data)
    with open(path, 'wb') as fh:
        fh.write(b''.join(out))


def _parse_ifd(get):
    """First IFD of a classic TIFF through ``get(offset, nbytes) -> bytes``: (byte order, {tag: values}) or None."""
    head = get(0, 8)
    bo = {b'II': '<', b'MM': '>'}.get(head[:2])
    if bo is None or len(head) < 8 or struct.unpack(bo + 'H', head[2:4])[0] != 42:
        return None
    (ifd,) = struct.unpack(bo + 'I', head[4:8])
    (n,) = struct.unpack(bo + 'H', get(ifd, 2))
    table = get(ifd + 2, 12 * n)
    tags = {}
    for i in range(n):
        e = 12 * i
        tag, typ, cnt = struct.unpack(bo + 'HHI', table[e:e + 8])
        if typ not in (1, 3, 4):
            continue
        sz = _SIZES[typ] * cnt
        if sz <= 4:
            raw = table[e + 8:e + 8 + sz]
        else:
            (off,) = struct.unpack(bo + 'I', table[e + 8:e + 12])
            raw = get(off, sz)
        tags[tag] = struct.unpack(bo + _TYPES[typ] * cnt, raw)
    return bo, tags


def _layout(bo, tags):
    """(h, w, samples per pixel, dtype, strip offsets, strip byte counts) of an uncompressed chunky 8/16-bit image, or None."""
    if tags.get(259, (1,))[0] != 1 or 273 not in tags:
        return None
    w, h = tags[256][0], tags[257][0]
    spp = tags.get(277, (1,))[0]
    bits = tags.get(258, (1,))[0]
    if bits not in (8, 16) or tags.get(339, (1,))[0] != 1 or tags.get(284, (1,))[0] != 1:
        return None
    dt = np.dtype(bo + ('u1' if bits == 8 else 'u2'))
    offsets = tags[273]
    counts = tags.get(279) or (h * w * spp * dt.itemsize,)
    if len(counts) != len(offsets):
        return None
    return h, w, spp, dt, offsets, counts


def _read_baseline_tiff(buf: bytes):
    parsed = _parse_ifd(lambda off, n: buf[off:off + n])
    lay = parsed and _layout(*parsed)
    if not lay:
        return None
    h, w, spp, dt, offsets, counts = lay
    parts = [buf[off:off + cnt] for off, cnt in zip(offsets, counts)]
    arr = np.frombuffer(b''.join(parts), dtype=dt, count=h * w * spp)
    arr = arr.astype(dt.newbyteorder('='))
    return arr.reshape((h, w) if spp == 1 else (h, w, spp))


def read_image_into(path: str, out: np.ndarray) -> bool:
    """Decode a tile file STRAIGHT INTO ``out`` (e.g. a slice of a page-locked staging buffer): the pixel bytes go from
    the file to their destination in one ``readinto`` -- no intermediate ``bytes``, no copy that holds the interpreter
    lock (``read_image`` makes four 8 MiB copies per 2048x2048 tile, most of them under the lock, which is what bounded
    the threaded ingest).  Handles what Squid writes: an uncompressed, chunky, native-endian 8/16-bit TIFF whose strips
    follow each other in the file, of exactly ``out``'s shape and dtype.  Returns False (``out`` untouched) for anything
    else; the caller falls back to ``read_image``."""
    if not path.lower().endswith(('.tif', '.tiff')) or not out.flags.c_contiguous or not out.flags.writeable:
        return False
    with open(path, 'rb', buffering=0) as fh:
        def get(off, n):
            fh.seek(off)
            return fh.read(n)
        try:
            parsed = _parse_ifd(get)
        except struct.error:
            return False
        lay = parsed and _layout(*parsed)
        if not lay:
            return False
        h, w, spp, dt, offsets, counts = lay
        shape = (h, w) if spp == 1 else (h, w, spp)
        if tuple(out.shape) != shape or out.dtype != dt.newbyteorder('=') or (dt.itemsize > 1 and not dt.isnative):
            return False
        if sum(counts) != out.nbytes or any(offsets[i] + counts[i] != offsets[i + 1] for i in range(len(offsets) - 1)):
            return False
        view = memoryview(out).cast('B')
        fh.seek(offsets[0])
        got = 0
        while got < len(view):
            k = fh.readinto(view[got:])
            if not k:
                raise ValueError(f"{path}: truncated TIFF ({got} of {len(view)} pixel bytes)")
            got += k
    return True


def read_image(path: str) -> np.ndarray:
    """Read one tile file as a numpy array (2-D, or HxWx3 for RGB)."""
    low = path.lower()
    if low.endswith(('.tif', '.tiff')):
        with open(path, 'rb') as fh:
            buf = fh.read()
        arr = _read_baseline_tiff(buf)
        if arr is not None:
            return arr
    try:
        from PIL import Image
    except ImportError as exc:  # pragma: no cover
        raise ValueError(f"cannot decode {path}: not a baseline TIFF and PIL is absent") from exc
    with Image.open(path) as im:
        return np.array(im)

"""Minimal OME-TIFF writer (uncompressed BigTIFF, one IFD per (t, c, z) plane, OME-XML in the
first ImageDescription) for the ``.ome.tiff`` output format.

The reference writes this format through aicsimageio's OmeTiffWriter (stitcher.py:742-765:
dim_order TCZYX, channel names, RGB channel colours, physical pixel sizes).  aicsimageio is not on
the hot path and absent offline; this is the package-free equivalent ("next" row 8(f)1).
"""
from __future__ import annotations

import struct
from typing import Sequence
from xml.sax.saxutils import escape

import numpy as np

_OME_TYPES = {np.dtype('uint8'): 'uint8', np.dtype('uint16'): 'uint16', np.dtype('float32'): 'float'}


def ome_xml(shape, dtype, channel_names: Sequence[str], channel_colors: Sequence[int], pixel_size_um: float,
            dz_um: float, name: str) -> str:
    t, c, z, y, x = (int(v) for v in shape)
    chans = []
    for i in range(c):
        label = escape(channel_names[i]) if i < len(channel_names) else f'Channel:{i}'
        rgb = channel_colors[i] if i < len(channel_colors) else 0xFFFFFF
        # OME colour is a signed 32-bit RGBA
        rgba = ((rgb & 0xFFFFFF) << 8) | 0xFF
        if rgba >= 2 ** 31:
            rgba -= 2 ** 32
        chans.append(f'<Channel ID="Channel:0:{i}" Name="{label}" SamplesPerPixel="1" Color="{rgba}"><LightPath/></Channel>')
    planes = ''.join(f'<TiffData FirstT="{ti}" FirstC="{ci}" FirstZ="{zi}" IFD="{(ti * c + ci) * z + zi}" PlaneCount="1"/>'
                     for ti in range(t) for ci in range(c) for zi in range(z))
    return ('<?xml version="1.0" encoding="UTF-8"?>'
            '<OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06" '
            'xmlns:xsi="http://www.w3.org/2001/XMLSchema-instance" '
            'xsi:schemaLocation="http://www.openmicroscopy.org/Schemas/OME/2016-06 '
            'http://www.openmicroscopy.org/Schemas/OME/2016-06/ome.xsd">'
            f'<Image ID="Image:0" Name="{escape(name)}">'
            f'<Pixels ID="Pixels:0" DimensionOrder="XYZCT" Type="{_OME_TYPES[np.dtype(dtype)]}" '
            f'SizeX="{x}" SizeY="{y}" SizeZ="{z}" SizeC="{c}" SizeT="{t}" '
            f'PhysicalSizeX="{pixel_size_um}" PhysicalSizeXUnit="µm" PhysicalSizeY="{pixel_size_um}" '
            f'PhysicalSizeYUnit="µm" PhysicalSizeZ="{dz_um}" PhysicalSizeZUnit="µm" BigEndian="false">'
            + ''.join(chans) + planes + '</Pixels></Image></OME>')


def write_ome_tiff(path: str, image: np.ndarray, *, pixel_size_um: float, dz_um: float = 1.0,
                   channel_names: Sequence[str] = (), channel_colors: Sequence[int] = (), name: str = 'stitched') -> str:
    """Write a (T, C, Z, Y, X) uint8/uint16/float32 array as an OME-TIFF (BigTIFF)."""
    if image.ndim != 5:
        raise ValueError(f"expected a 5-D TCZYX array, got {image.shape}")
    dt = np.dtype(image.dtype)
    if dt not in _OME_TYPES:
        raise ValueError(f"unsupported dtype {dt}")
    t, c, z, y, x = image.shape
    desc = ome_xml(image.shape, dt, channel_names, channel_colors, pixel_size_um, dz_um, name).encode('utf-8') + b'\0'
    plane_bytes = y * x * dt.itemsize
    n_planes = t * c * z
    sample_format = 3 if dt.kind == 'f' else 1

    def ifd(strip_off, desc_off, next_off):
        ents = [(256, 16, 1, x), (257, 16, 1, y), (258, 3, 1, dt.itemsize * 8), (259, 3, 1, 1), (262, 3, 1, 1)]
        if desc_off is not None:
            ents.append((270, 2, len(desc), desc_off))
        ents += [(273, 16, 1, strip_off), (277, 3, 1, 1), (278, 16, 1, y), (279, 16, 1, plane_bytes),
                 (339, 3, 1, sample_format)]
        ents.sort()
        out = [struct.pack('<Q', len(ents))]
        for tag, typ, cnt, val in ents:
            out.append(struct.pack('<HHQQ', tag, typ, cnt, val))
        out.append(struct.pack('<Q', next_off))
        return b''.join(out)

    ifd_size0 = 8 + 11 * 20 + 8
    ifd_size = 8 + 10 * 20 + 8
    # layout: header(16) | description | IFD0 | IFD1.. | planes
    desc_off = 16
    ifd0_off = (desc_off + len(desc) + 7) & ~7
    data_off = ifd0_off + ifd_size0 + (n_planes - 1) * ifd_size
    data_off = (data_off + 15) & ~15
    with open(path, 'wb') as fh:
        fh.write(struct.pack('<2sHHHQ', b'II', 43, 8, 0, ifd0_off))
        fh.write(desc)
        fh.write(b'\0' * (ifd0_off - desc_off - len(desc)))
        off = ifd0_off
        for p in range(n_planes):
            size = ifd_size0 if p == 0 else ifd_size
            nxt = off + size if p + 1 < n_planes else 0
            fh.write(ifd(data_off + p * plane_bytes, desc_off if p == 0 else None, nxt))
            off += size
        fh.write(b'\0' * (data_off - off))
        flat = image.reshape(n_planes, y, x)
        for p in range(n_planes):
            fh.write(np.ascontiguousarray(flat[p]).astype(dt.newbyteorder('<'), copy=False).tobytes())
    return path


def read_ome_tiff(path: str):
    """Planes and OME-XML back from a file written by write_ome_tiff (tests)."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    bo, ver, osz, _, off = struct.unpack('<2sHHHQ', buf[:16])
    assert bo == b'II' and ver == 43 and osz == 8
    planes, xml = [], None
    while off:
        (n,) = struct.unpack('<Q', buf[off:off + 8])
        tags = {}
        for i in range(n):
            tag, typ, cnt, val = struct.unpack('<HHQQ', buf[off + 8 + 20 * i: off + 28 + 20 * i])
            tags[tag] = (typ, cnt, val)
        w, h, bits = tags[256][2], tags[257][2], tags[258][2]
        fmt = tags.get(339, (3, 1, 1))[2]
        dt = np.dtype('<f4') if fmt == 3 else np.dtype(f'<u{bits // 8}')
        planes.append(np.frombuffer(buf, dtype=dt, count=w * h, offset=tags[273][2]).reshape(h, w))
        if 270 in tags:
            xml = buf[tags[270][2]:tags[270][2] + tags[270][1] - 1].decode('utf-8')
        (off,) = struct.unpack('<Q', buf[off + 8 + 20 * n: off + 16 + 20 * n])
    return planes, xml

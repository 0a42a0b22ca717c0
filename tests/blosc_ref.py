"""Independent reader of Blosc-1 frames and of zarr-v2 / NGFF-0.4 stores, written from the format documents
(c-blosc 1.x README_HEADER.rst / README_CHUNK_FORMAT.rst, the LZ4 block format description, the zarr v2 spec and
the OME-NGFF 0.4 spec) -- NOT from this repository's writer or reader (image-stitcher_amd/omezarr.py).  The `blosc`,
`numcodecs` and `zarr` packages are absent offline; this stands in for them as the third-party side of the check.
"""
import json
import os
import struct

import numpy as np

BLOSC_MIN_BUFFERSIZE = 128       # c-blosc: blocks are split into `typesize` streams only above this many bytes each
BLOSC_MAX_SPLITS = 16


def lz4_decompress_block(data: memoryview, expected: int) -> bytes:
    """LZ4 block format: sequences of token | [literal length bytes] | literals | offset (LE16) | [match length bytes];
    the last sequence ends after its literals."""
    dst = bytearray(expected)
    ip, op, n = 0, 0, len(data)
    while True:
        token = data[ip]
        ip += 1
        run = token >> 4
        if run == 15:
            while True:
                s = data[ip]
                ip += 1
                run += s
                if s < 255:
                    break
        if op + run > expected or ip + run > n:
            raise ValueError("LZ4: literal run overflows")
        dst[op:op + run] = data[ip:ip + run]
        ip += run
        op += run
        if ip == n:
            break                       # end of block: the last sequence has no match part
        (offset,) = struct.unpack_from('<H', data, ip)
        ip += 2
        if offset == 0 or offset > op:
            raise ValueError("LZ4: bad offset")
        mlen = (token & 15)
        if mlen == 15:
            while True:
                s = data[ip]
                ip += 1
                mlen += s
                if s < 255:
                    break
        mlen += 4
        if op + mlen > expected:
            raise ValueError("LZ4: match overflows")
        ref = op - offset
        while mlen > 0:                 # overlapping copies repeat the last `offset` bytes
            step = min(mlen, op - ref)
            dst[op:op + step] = dst[ref:ref + step]
            op += step
            ref += step
            mlen -= step
    if op != expected:
        raise ValueError(f"LZ4: block ends at {op}, expected {expected}")
    return bytes(dst)


def parse_header(frame) -> dict:
    version, versionlz, flags, typesize, nbytes, blocksize, cbytes = struct.unpack_from('<BBBBIII', frame, 0)
    return dict(version=version, versionlz=versionlz, flags=flags, typesize=typesize, nbytes=nbytes, blocksize=blocksize,
                cbytes=cbytes, shuffle=bool(flags & 0x01), memcpyed=bool(flags & 0x02), bitshuffle=bool(flags & 0x04),
                dont_split=bool(flags & 0x10), codec=flags >> 5)


def blosc_decompress(frame: bytes) -> bytes:
    mv = memoryview(frame)
    h = parse_header(mv)
    if h['version'] != 2:
        raise ValueError(f"Blosc format version {h['version']}")
    if h['cbytes'] != len(frame):
        raise ValueError(f"header says {h['cbytes']} compressed bytes, the chunk has {len(frame)}")
    nbytes, typesize, blocksize = h['nbytes'], h['typesize'], h['blocksize']
    if h['memcpyed']:
        return bytes(mv[16:16 + nbytes])
    if h['codec'] != 1 or h['versionlz'] != 1:
        raise ValueError("not an LZ4 frame")
    if h['bitshuffle']:
        raise ValueError("bit shuffle is not read here")
    nblocks = (nbytes + blocksize - 1) // blocksize if nbytes else 0
    bstarts = struct.unpack_from(f'<{nblocks}i', mv, 16)
    parts = []
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        split = (not h['dont_split']) and typesize <= BLOSC_MAX_SPLITS and blocksize // typesize >= BLOSC_MIN_BUFFERSIZE and not leftover
        nsplits = typesize if split else 1
        neblock = bsize // nsplits
        at = bstarts[b]
        streams = []
        for _ in range(nsplits):
            (cb,) = struct.unpack_from('<i', mv, at)
            at += 4
            streams.append(bytes(mv[at:at + cb]) if cb == neblock else lz4_decompress_block(mv[at:at + cb], neblock))
            at += cb
        block = b''.join(streams)
        if h['shuffle'] and typesize > 1:
            nel = bsize // typesize
            body = np.frombuffer(block, dtype=np.uint8, count=nel * typesize).reshape(typesize, nel)
            block = np.ascontiguousarray(body.T).tobytes() + block[nel * typesize:]
        parts.append(block)
    out = b''.join(parts)
    if len(out) != nbytes:
        raise ValueError("decoded size differs from the header's nbytes")
    return out


def read_zarr_v2_array(path: str):
    """(array, .zarray metadata) of one zarr-v2 array directory: chunk keys joined by `dimension_separator`, missing
    chunks = fill_value, edge chunks stored at full chunk size, C order, little-endian dtype strings."""
    with open(os.path.join(path, '.zarray')) as fh:
        meta = json.load(fh)
    assert meta['zarr_format'] == 2 and meta['order'] == 'C' and meta['filters'] is None
    shape, chunks = tuple(meta['shape']), tuple(meta['chunks'])
    dtype = np.dtype(meta['dtype'])
    sep = meta.get('dimension_separator', '.')
    comp = meta['compressor']
    out = np.full(shape, meta['fill_value'], dtype=dtype)
    grid = [-(-s // c) for s, c in zip(shape, chunks)]
    seen = 0
    for idx in np.ndindex(*grid):
        key = sep.join(str(i) for i in idx)
        p = os.path.join(path, *key.split('/')) if sep == '/' else os.path.join(path, key)
        if not os.path.isfile(p):
            continue
        seen += 1
        with open(p, 'rb') as fh:
            raw = fh.read()
        if comp is None:
            data = raw
        elif comp['id'] == 'blosc':
            data = blosc_decompress(raw)
        elif comp['id'] == 'zlib':
            import zlib
            data = zlib.decompress(raw)
        else:
            raise ValueError(f"codec {comp['id']}")
        block = np.frombuffer(data, dtype=dtype)
        assert block.size == int(np.prod(chunks)), f"chunk {key}: {block.size} elements, a full chunk has {int(np.prod(chunks))}"
        block = block.reshape(chunks)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
    return out, meta, seen


def check_ngff_group(path: str, n_levels: int, shape5, pixel_size_um: float, dz_um: float, channel_names):
    """NGFF 0.4 multiscales + omero metadata of an image group, as the reference writes them (stitcher.py:801-856)."""
    with open(os.path.join(path, '.zgroup')) as fh:
        assert json.load(fh) == {'zarr_format': 2}
    with open(os.path.join(path, '.zattrs')) as fh:
        attrs = json.load(fh)
    (ms,) = attrs['multiscales']
    assert ms['version'] == '0.4'
    assert [a['name'] for a in ms['axes']] == ['t', 'c', 'z', 'y', 'x']
    assert [a['type'] for a in ms['axes']] == ['time', 'channel', 'space', 'space', 'space']
    assert [a.get('unit') for a in ms['axes']] == ['second', None, 'micrometer', 'micrometer', 'micrometer']
    assert [d['path'] for d in ms['datasets']] == [str(i) for i in range(n_levels)]
    for lv, d in enumerate(ms['datasets']):
        (tr,) = d['coordinateTransformations']
        assert tr['type'] == 'scale'
        np.testing.assert_allclose(tr['scale'], [1, 1, dz_um, pixel_size_um * 2 ** lv, pixel_size_um * 2 ** lv])
        with open(os.path.join(path, str(lv), '.zarray')) as fh:
            za = json.load(fh)
        want = list(shape5[:3]) + [shape5[3] >> lv, shape5[4] >> lv]
        assert za['shape'] == want and za['chunks'][:3] == [1, 1, 1]
    assert [c['label'] for c in attrs['omero']['channels']] == list(channel_names)
    return attrs

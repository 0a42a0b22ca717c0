"""The two Blosc-1 / LZ4 readers (the store reader in omezarr.py and the independent one in tests/blosc_ref.py) on frames
built here by hand from the format documents: known LZ4 vectors, split and unsplit blocks, shuffle, memcpy'd frames."""
import struct

import numpy as np
import pytest

import blosc_ref
from image_stitcher_amd import omezarr


def lz4_greedy(data: bytes) -> bytes:
    """A plain greedy LZ4 block encoder (reference for the tests; quadratic search, tiny inputs only)."""
    n, out, anchor, i = len(data), bytearray(), 0, 0

    def emit(lit, off=None, ml=0):
        token = (min(len(lit), 15) << 4) | (min(ml - 4, 15) if off else 0)
        out.append(token)
        if len(lit) >= 15:
            r = len(lit) - 15
            out.extend([255] * (r // 255) + [r % 255])
        out.extend(lit)
        if off:
            out.extend(struct.pack('<H', off))
            if ml - 4 >= 15:
                r = ml - 4 - 15
                out.extend([255] * (r // 255) + [r % 255])

    while i <= n - 12:
        best = None
        for j in range(max(0, i - 300), i):
            if data[j:j + 4] == data[i:i + 4]:
                best = j
        if best is None:
            i += 1
            continue
        ml = 4
        while i + ml < n - 5 and data[best + ml] == data[i + ml]:
            ml += 1
        emit(data[anchor:i], i - best, ml)
        i += ml
        anchor = i
    emit(data[anchor:])
    return bytes(out)


def frame(raw: bytes, typesize: int, blocksize: int, shuffle: bool, dont_split: bool) -> bytes:
    nblocks = -(-len(raw) // blocksize)
    body, bstarts = bytearray(), []
    for b in range(nblocks):
        blk = raw[b * blocksize:(b + 1) * blocksize]
        if shuffle and typesize > 1:
            nel = len(blk) // typesize
            blk = np.frombuffer(blk, np.uint8, nel * typesize).reshape(nel, typesize).T.tobytes() + blk[nel * typesize:]
        split = (not dont_split) and blocksize // typesize >= 128 and len(blk) == blocksize
        bstarts.append(16 + 4 * nblocks + len(body))
        ns = typesize if split else 1
        for k in range(ns):
            part = blk[k * len(blk) // ns:(k + 1) * len(blk) // ns]
            c = lz4_greedy(part)
            if len(c) >= len(part):
                c = part
            body += struct.pack('<i', len(c)) + c
    flags = (1 if shuffle and typesize > 1 else 0) | (0x10 if dont_split else 0) | (1 << 5)
    total = 16 + 4 * nblocks + len(body)
    return struct.pack('<BBBBIII', 2, 1, flags, typesize, len(raw), blocksize, total) + struct.pack(f'<{nblocks}i', *bstarts) + bytes(body)


def test_known_lz4_vector():
    # 20 x 'a': literal 'a', match offset 1 length 14, then the mandatory 5 trailing literals
    block = bytes([0x1A, ord('a'), 0x01, 0x00, 0x50]) + b'aaaaa'
    assert blosc_ref.lz4_decompress_block(memoryview(block), 20) == b'a' * 20
    assert bytes(omezarr._lz4_block(block, 20)) == b'a' * 20
    assert lz4_greedy(b'a' * 20) == block
    with pytest.raises(ValueError):
        blosc_ref.lz4_decompress_block(memoryview(block), 19)


@pytest.mark.parametrize('typesize', [1, 2])
@pytest.mark.parametrize('dont_split', [True, False])
def test_both_readers_on_hand_built_frames(typesize, dont_split):
    rng = np.random.default_rng(typesize * 2 + dont_split)
    smooth = (np.arange(1500) // 7 % 251).astype(np.uint16 if typesize == 2 else np.uint8)
    noisy = rng.integers(0, 1 << (8 * typesize), 700).astype(smooth.dtype)
    raw = np.concatenate([smooth, noisy, np.zeros(600, smooth.dtype), smooth[:123]]).tobytes()
    for blocksize in (512, 1024, len(raw)):
        f = frame(raw, typesize, blocksize, shuffle=True, dont_split=dont_split)
        assert blosc_ref.blosc_decompress(f) == raw and omezarr.blosc_decode(f) == raw
        h = blosc_ref.parse_header(f)
        assert (h['nbytes'], h['cbytes'], h['typesize'], h['codec'], h['dont_split']) == (len(raw), len(f), typesize, 1, dont_split)
    # memcpy'd frame
    f = struct.pack('<BBBBIII', 2, 1, 0x02, typesize, len(raw), len(raw), 16 + len(raw)) + raw
    assert blosc_ref.blosc_decompress(f) == raw and omezarr.blosc_decode(f) == raw
    with pytest.raises(ValueError):
        blosc_ref.blosc_decompress(f[:-1])


def test_host_chunk_writers_refuse_codecs_they_do_not_implement(tmp_path):
    """emit_chunk / write_plane_levels used to fall through to zlib for anything that was not 'none' --
    'blosc' (the Stitcher default, encoded on the DEVICE) then left zlib bytes under a .zarray that says blosc."""
    from image_stitcher_amd import omezarr
    plane = np.arange(64 * 64, dtype=np.uint16).reshape(1, 64, 64)
    for bad in ('blosc', 'lz4', 'zstd'):
        with pytest.raises(ValueError, match='device encoder'):
            omezarr.write_plane_levels(str(tmp_path / 's'), [plane], [(0, 0, 0)], (1, 1, 1, 32, 32), compression=bad)
        with pytest.raises(ValueError, match='device encoder'):
            omezarr.emit_chunk(str(tmp_path / 's'), (0, 0, 0, 0, 0, 0, plane[0], 0, 0, 32, 32), bad)
    assert not (tmp_path / 's').exists()
    for ok in ('zlib', 'none', None):
        assert omezarr.write_plane_levels(str(tmp_path / f's_{ok}'), [plane], [(0, 0, 0)], (1, 1, 1, 32, 32), compression=ok) > 0


def test_both_readers_decode_genuine_c_blosc_frames():
    """frames produced by the GENUINE c-blosc 1.21.0 (imagecodecs 2021.8.26 under the authoring
    container's /opt/conda/bin/python3.9; tests/golden/make_blosc_golden.py cblosc) -- LZ4 with and without byte
    shuffle, levels 1 / 5 / 9, typesize 1 / 2, split streams, an incompressible (memcpy'd) chunk, a 42-byte chunk --
    decode to the seeded raw chunk with BOTH readers of this repo.  That pins the readers the device encoder's tests
    lean on to third-party output."""
    import json
    import os
    import sys
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, golden)
    import make_blosc_golden as G
    with open(os.path.join(golden, 'blosc_cblosc_frames.json')) as fh:
        index = json.load(fh)
    assert index['encoder'] == 'c-blosc 1.21.0'
    frames = np.load(os.path.join(golden, 'blosc_cblosc_frames.npz'))
    seen_flags = set()
    for e in index['frames']:
        frame = frames[e['name']].tobytes()
        raw = G.chunk(e['kind'], e['h'], e['w'], e['dtype'], e['seed']).tobytes()
        assert G.sha(raw) == e['raw_sha256'] and len(frame) == e['cbytes']       # the seeded chunk is the one c-blosc saw
        hd = blosc_ref.parse_header(frame)
        assert hd['nbytes'] == len(raw) and hd['cbytes'] == len(frame) and hd['typesize'] == np.dtype(e['dtype']).itemsize
        assert blosc_ref.blosc_decompress(frame) == raw, e['name']
        assert omezarr.blosc_decode(frame) == raw, e['name']
        seen_flags.add(e['flags'])
    assert {0x21, 0x20, 0x23} <= seen_flags          # shuffled, unshuffled, memcpy'd -- split (no 0x10) streams throughout
    assert len(index['frames']) >= 15

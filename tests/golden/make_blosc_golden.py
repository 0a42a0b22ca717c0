#!/usr/bin/env python3
"""Pins the Blosc-1 / LZ4 chunk codec (the reference's store gets zarr's default compressor,
stitcher.py:814-818) against the GENUINE c-blosc, in both directions.  c-blosc 1.21.0 is in the authoring container
inside imagecodecs 2021.8.26 under /opt/conda/bin/python3.9; the GPU box has neither, so the device frames travel
back as data.

    /opt/conda/bin/python3.9 tests/golden/make_blosc_golden.py cblosc
        genuine c-blosc frames of seeded chunks (LZ4, shuffle on/off, levels 1/5/9, typesize 1/2, split and unsplit
        blocks, a memcpy'd frame) -> tests/golden/blosc_cblosc_frames.npz.  tests/test_blosc_cpu.py: both repo
        DECODERS (omezarr.blosc_decode, tests/blosc_ref.py) reproduce the raw bytes -> the readers are pinned.

    python3 tests/golden/make_blosc_golden.py dump gpurun_out/r3/blosc_device_frames.npz        (on the GPU box)
        the device ENCODER (csrc/blosc.hip) on the seeded chunks of ``device_cases()`` -> every frame + its raw chunk.

    /opt/conda/bin/python3.9 tests/golden/make_blosc_golden.py verify gpurun_out/r3/blosc_device_frames.npz
        every dumped device frame decoded by c-blosc 1.21.0 == its raw chunk -> tests/golden/blosc_device_frames.json
        {case: sha256(frame), sha256(raw)}.  tests/test_blosc_gpu.py re-encodes the same seeded chunks on the device
        and checks the digests against that table (the encoder is deterministic), so a frame c-blosc has decoded is the
        frame the product writes.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def chunk(kind: str, h: int, w: int, dtype: str, seed: int) -> np.ndarray:
    """Seeded chunk contents from integer arithmetic only (identical under numpy 1.26 and 2.x)."""
    info = np.iinfo(dtype)
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    lcg = (y * 1103515245 + x * 12345 + seed * 2654435761 + (y * x) * 40503) & 0xFFFFFFFF
    noise = (lcg * 1664525 + 1013904223) >> 11 & 0xFFFF
    if kind == 'smooth':      # microscope-like: slowly varying signal, a few counts of noise in the low bits
        v = 2000 + (y * 3 + x * 2 + seed) // 4 + (noise & 7)
    elif kind == 'noise':     # incompressible: blocks go out raw
        v = noise
    elif kind == 'ramp':      # long exact matches, overlapping copies
        v = (x // 9) % 200 + seed
    elif kind == 'blocks':    # 8 x 8 constant blocks
        v = ((y // 8) * 131 + (x // 8) * 31 + seed) * 257
    elif kind == 'sparse':    # mostly zero with isolated bright pixels
        v = np.where((noise & 0x3FF) == 0, noise, 0)
    elif kind == 'constant':
        v = np.full((h, w), 777 + seed, dtype=np.int64)
    elif kind == 'zero':
        v = np.zeros((h, w), dtype=np.int64)
    else:
        raise ValueError(kind)
    return (v % (info.max + 1)).astype(dtype)


def device_cases():
    """(name, plane [h, w], chunk_y, chunk_x): what the device encoder is run on.  Planes larger than one chunk give
    edge-padded chunks (zarr pads edge chunks with zeros); every chunk of every plane is one frame."""
    out = []
    for kind in ('smooth', 'noise', 'ramp', 'blocks', 'sparse', 'constant'):
        out.append((f'{kind}_u16_512', chunk(kind, 512, 512, 'uint16', 3), 512, 512))
    for kind in ('smooth', 'noise', 'ramp', 'sparse'):
        out.append((f'{kind}_u8_512', chunk(kind, 512, 512, 'uint8', 5), 512, 512))
    out.append(('smooth_u16_edge', chunk('smooth', 700, 900, 'uint16', 7), 512, 512))       # 4 chunks, 3 edge-padded
    out.append(('blocks_u16_small_chunks', chunk('blocks', 200, 260, 'uint16', 9), 128, 128))  # 6 chunks, blocks < 16 KiB
    out.append(('smooth_u8_edge', chunk('smooth', 300, 520, 'uint8', 11), 256, 512))
    out.append(('half_zero_u16', np.concatenate([chunk('smooth', 512, 512, 'uint16', 13), chunk('zero', 512, 512, 'uint16', 0)]),
                512, 512))                                                                      # second chunk: no frame
    return out


def padded_chunks(plane, cy, cx):
    h, w = plane.shape
    for iy in range(-(-h // cy)):
        for ix in range(-(-w // cx)):
            full = np.zeros((cy, cx), dtype=plane.dtype)
            part = plane[iy * cy:(iy + 1) * cy, ix * cx:(ix + 1) * cx]
            full[:part.shape[0], :part.shape[1]] = part
            yield iy, ix, full


def make_cblosc(path):
    import imagecodecs
    assert imagecodecs.blosc_version() == 'c-blosc 1.21.0', imagecodecs.blosc_version()
    arrays, index = {}, []
    cases = [   # kind, h, w, dtype, level, shuffle, blocksize (0 = c-blosc's automatic choice)
        ('smooth', 128, 128, 'uint16', 5, 1, 0), ('smooth', 128, 128, 'uint16', 1, 1, 0), ('smooth', 128, 128, 'uint16', 9, 1, 0),
        ('smooth', 128, 128, 'uint16', 5, 0, 0), ('noise', 128, 128, 'uint16', 5, 1, 0), ('ramp', 128, 128, 'uint16', 5, 1, 0),
        ('blocks', 128, 128, 'uint16', 5, 1, 0), ('sparse', 128, 128, 'uint16', 5, 1, 0), ('constant', 64, 64, 'uint16', 5, 1, 0),
        ('smooth', 128, 128, 'uint8', 5, 1, 0), ('ramp', 128, 128, 'uint8', 5, 0, 0), ('noise', 64, 64, 'uint8', 9, 1, 0),
        ('smooth', 128, 128, 'uint16', 5, 1, 4096), ('smooth', 128, 128, 'uint16', 5, 1, 16384), ('ramp', 100, 77, 'uint16', 5, 1, 0),
        ('smooth', 512, 512, 'uint16', 5, 1, 0),          # one chunk of the Stitcher's default geometry, zarr's defaults
        ('smooth', 7, 3, 'uint16', 5, 1, 0),              # smaller than any block: c-blosc stores it as it sees fit
    ]
    for k, (kind, h, w, dtype, level, shuffle, blocksize) in enumerate(cases):
        raw = chunk(kind, h, w, dtype, 100 + k).tobytes()
        kw = dict(level=level, compressor='lz4', typesize=np.dtype(dtype).itemsize, shuffle=shuffle)
        if blocksize:
            kw['blocksize'] = blocksize
        frame = imagecodecs.blosc_encode(raw, **kw)
        assert imagecodecs.blosc_decode(frame) == raw
        name = f'c{k:02d}_{kind}_{h}x{w}_{dtype}_l{level}_s{shuffle}_b{blocksize}'
        arrays[name] = np.frombuffer(frame, dtype=np.uint8)
        index.append(dict(name=name, kind=kind, h=h, w=w, dtype=dtype, seed=100 + k, level=level, shuffle=shuffle,
                          blocksize=blocksize, nbytes=len(raw), cbytes=len(frame), raw_sha256=sha(raw), flags=int(frame[2])))
    np.savez_compressed(path, **arrays)
    with open(os.path.splitext(path)[0] + '.json', 'w') as fh:
        json.dump({'encoder': imagecodecs.blosc_version(), 'imagecodecs': imagecodecs.__version__, 'frames': index}, fh, indent=1)
    print(f'{len(index)} genuine c-blosc frames, {sum(a.nbytes for a in arrays.values())} bytes -> {path}')


def dump_device(path):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from image_stitcher_amd import native
    dev = torch.device('cuda:0')
    arrays, n = {}, 0
    for name, plane, cy, cx in device_cases():
        buf = native.blosc_encode_planes(torch.from_numpy(plane[None]).to(dev), cy, cx)
        torch.cuda.synchronize()
        assert int(buf.status.item()) == 0
        off, out = buf.offsets.cpu().numpy(), buf.out.cpu().numpy()
        ncx = -(-plane.shape[1] // cx)
        for iy, ix, full in padded_chunks(plane, cy, cx):
            i = iy * ncx + ix
            arrays[f'{name}__{iy}_{ix}__frame'] = out[off[i]:off[i + 1]].copy()
            arrays[f'{name}__{iy}_{ix}__raw'] = np.frombuffer(full.tobytes(), dtype=np.uint8)
            n += 1
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, **arrays)
    print(f'{n} device frames -> {path}')


def verify_device(path):
    import imagecodecs
    assert imagecodecs.blosc_version() == 'c-blosc 1.21.0', imagecodecs.blosc_version()
    z = np.load(path)
    table, n_frames = {}, 0
    for key in sorted(k for k in z.files if k.endswith('__frame')):
        name = key[:-len('__frame')]
        frame, raw = z[key].tobytes(), z[name + '__raw'].tobytes()
        if not frame:
            assert not any(raw), f'{name}: no frame for a chunk that is not all zero'
            table[name] = dict(frame_sha256=None, raw_sha256=sha(raw), nbytes=len(raw), cbytes=0)
            continue
        got = imagecodecs.blosc_decode(frame)          # the genuine c-blosc 1.21.0 decompressor
        assert got == raw, f'{name}: c-blosc decodes the device frame to something else'
        table[name] = dict(frame_sha256=sha(frame), raw_sha256=sha(raw), nbytes=len(raw), cbytes=len(frame))
        n_frames += 1
    out = os.path.join(HERE, 'blosc_device_frames.json')
    with open(out, 'w') as fh:
        json.dump({'decoded_by': imagecodecs.blosc_version(), 'imagecodecs': imagecodecs.__version__,
                   'note': 'frames produced by csrc/blosc.hip on an MI355X (make_blosc_golden.py dump), every one decoded by the '
                           'genuine c-blosc to the raw chunk; test_blosc_gpu.py re-encodes and compares the digests',
                   'frames': table}, fh, indent=1)
    print(f'{n_frames} device frames decoded by {imagecodecs.blosc_version()} (+ {len(table) - n_frames} all-zero chunks without a '
          f'frame) -> {out}')


if __name__ == '__main__':
    cmd = sys.argv[1] if len(sys.argv) > 1 else ''
    if cmd == 'cblosc':
        make_cblosc(os.path.join(HERE, 'blosc_cblosc_frames.npz'))
    elif cmd == 'dump':
        dump_device(sys.argv[2])
    elif cmd == 'verify':
        verify_device(sys.argv[2])
    else:
        raise SystemExit(__doc__)

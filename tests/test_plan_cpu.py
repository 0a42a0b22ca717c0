"""Host-side pieces that need no GPU: the C-ABI loads and exports every declared symbol, and
the span planner (plan.cpp) partitions the canvas so that replaying its table in numpy
reproduces the oracle's fusion exactly."""
import re
import os

import numpy as np
import pytest

from image_stitcher_amd import native
from oracle import stitch_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HDR = np.dtype([('magic', '<u4'), ('mode', '<i4'), ('canvas_h', '<i4'), ('canvas_w', '<i4'), ('tile_h', '<i4'),
                ('tile_w', '<i4'), ('n_tiles', '<i4'), ('max_refs', '<i4'), ('n_spans', '<i8'), ('n_refs', '<i8'),
                ('n_items', '<i8'), ('off_spans', '<i8'), ('off_refs', '<i8'), ('off_items', '<i8'),
                ('covered', '<i8'), ('reserved', '<i8')])
SPAN = np.dtype([('dst_y', '<i4'), ('dst_x', '<i4'), ('h', '<i4'), ('w', '<i4'), ('nref', '<i4'), ('ref0', '<i4'),
                 ('pad', '<i4', (2,))])
REF = np.dtype([('tile', '<i4'), ('src_y', '<i4'), ('src_x', '<i4'), ('pad', '<i4')])
ITEM = np.dtype([('dst_y', '<i4'), ('dst_x', '<i4'), ('hw', '<i4'), ('nref', '<i4'), ('a', '<i4'), ('b', '<i4'),
                 ('c', '<i4'), ('span', '<i4')])


def decode(plan):
    t = plan.table
    hd = t[:HDR.itemsize].view(HDR)[0]
    assert hd['magic'] == 0x53514654
    spans = t[hd['off_spans']:hd['off_spans'] + hd['n_spans'] * SPAN.itemsize].view(SPAN)
    refs = t[hd['off_refs']:hd['off_refs'] + hd['n_refs'] * REF.itemsize].view(REF)
    items = t[hd['off_items']:hd['off_items'] + hd['n_items'] * ITEM.itemsize].view(ITEM)
    return hd, spans, refs, items


def replay_overwrite(plan, tiles, flat=None):
    hd, spans, refs, items = decode(plan)
    out = np.full((plan.canvas_h, plan.canvas_w), 0xAAAA, dtype=tiles.dtype)   # poison: every voxel must be written
    hits = np.zeros((plan.canvas_h, plan.canvas_w), dtype=np.int32)
    for it in items:
        h, w = it['hw'] >> 16, it['hw'] & 0xFFFF
        assert 0 < h <= 8 and 0 < w <= 2048
        ys, xs = it['dst_y'], it['dst_x']
        sp = spans[it['span']]
        assert sp['dst_y'] <= ys and ys + h <= sp['dst_y'] + sp['h'] and sp['dst_x'] <= xs and xs + w <= sp['dst_x'] + sp['w']
        hits[ys:ys + h, xs:xs + w] += 1
        if it['nref'] == 0:
            out[ys:ys + h, xs:xs + w] = 0
            continue
        sy, sx = it['b'], it['c']
        src = tiles[it['a'], sy:sy + h, sx:sx + w]
        if flat is not None:
            src = O.apply_flatfield(src, flat[sy:sy + h, sx:sx + w], tiles.dtype.type)
        out[ys:ys + h, xs:xs + w] = src
    assert (hits == 1).all(), "spans/items must tile the canvas exactly once"
    return out


def random_rects(rng, n, th, tw, ch, cw, crop=True):
    rects = []
    for _ in range(n):
        sy, sx = (int(rng.integers(0, th // 3)), int(rng.integers(0, tw // 3))) if crop else (0, 0)
        h = int(rng.integers(1, th - sy + 1))
        w = int(rng.integers(1, tw - sx + 1))
        rects.append((sy, sx, h, w, int(rng.integers(0, ch)), int(rng.integers(0, cw))))
    return np.array(rects)


def test_library_exports_every_declared_symbol():
    L = native.lib()
    with open(os.path.join(ROOT, 'include', 'squidstitch.h')) as fh:
        hdr = fh.read()
    declared = set(re.findall(r'\b(sq_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.sq_version() == 102


def test_struct_layouts_match_header():
    assert native.RECT_DTYPE.itemsize == 24
    assert native.PAIR_DTYPE.itemsize == 24
    assert native.RESULT_DTYPE.itemsize == 48
    assert native.SYNTH_DTYPE.itemsize == 32


@pytest.mark.parametrize('seed', range(6))
def test_overwrite_plan_replays_to_oracle(seed):
    rng = np.random.default_rng(seed)
    th, tw = 37, 53
    ch, cw = int(rng.integers(40, 200)), int(rng.integers(40, 200))
    n = int(rng.integers(1, 25))
    rects = random_rects(rng, n, th, tw, ch, cw)
    tiles = rng.integers(1, 65535, size=(n, th, tw), dtype=np.uint16)
    plan = native.FusePlan(rects, th, tw, ch, cw, native.SQ_FUSE_OVERWRITE)
    want = O.fuse_plane_overwrite(list(tiles), rects, ch, cw)
    np.testing.assert_array_equal(replay_overwrite(plan, tiles), want)
    assert plan.covered_voxels == int((want != 0).sum())
    assert plan.max_refs <= 1


def test_overwrite_plan_with_flatfield_and_grid():
    rng = np.random.default_rng(99)
    th, tw = 64, 80
    rows, cols = 3, 4
    rects = []
    for r in range(rows):
        for c in range(cols):
            top = 9 if r else 0
            left = 11 if c else 0
            bot = 9 if r < rows - 1 else 0
            right = 11 if c < cols - 1 else 0
            rects.append((top, left, th - top - bot, tw - left - right, r * (th - 18) + c * 3 + top,
                          c * (tw - 22) + (rows - 1 - r) * 2 + left))
    rects = np.array(rects)
    ch, cw = th + (rows - 1) * (th + 18) + 9, tw + (cols - 1) * (tw - 22) + 4   # oversize height like the reference
    tiles = rng.integers(0, 65536, size=(len(rects), th, tw), dtype=np.uint16)
    flat = (0.5 + rng.random((th, tw))).astype(np.float32)
    plan = native.FusePlan(rects, th, tw, ch, cw)
    want = O.fuse_plane_overwrite(list(tiles), rects, ch, cw, flat)
    np.testing.assert_array_equal(replay_overwrite(plan, tiles, flat), want)


def test_feather_plan_lists_all_covering_tiles_in_order():
    rng = np.random.default_rng(5)
    th, tw, ch, cw = 20, 24, 50, 60
    rects = random_rects(rng, 6, th, tw, ch - 10, cw - 10, crop=False)
    plan = native.FusePlan(rects, th, tw, ch, cw, native.SQ_FUSE_FEATHER)
    hd, spans, refs, items = decode(plan)
    cover = [[[] for _ in range(cw)] for _ in range(ch)]
    for i, (sy, sx, h, w, dy, dx) in enumerate(O.clip_rects(rects, ch, cw)):
        for y in range(dy, dy + h):
            for x in range(dx, dx + w):
                cover[y][x].append((i, sy + y - dy, sx + x - dx))
    seen = np.zeros((ch, cw), dtype=int)
    for sp in spans:
        for y in range(sp['h']):
            for x in range(sp['w']):
                got = [(refs[sp['ref0'] + k]['tile'], refs[sp['ref0'] + k]['src_y'] + y, refs[sp['ref0'] + k]['src_x'] + x)
                       for k in range(sp['nref'])]
                assert got == cover[sp['dst_y'] + y][sp['dst_x'] + x]
                seen[sp['dst_y'] + y, sp['dst_x'] + x] += 1
    assert (seen == 1).all()


def test_plan_rejects_bad_input():
    with pytest.raises(native.NativeError, match='outside'):
        native.FusePlan(np.array([(0, 0, 40, 10, 0, 0)]), 32, 32, 64, 64)
    with pytest.raises(native.NativeError, match='negative'):
        native.FusePlan(np.array([(0, 0, 10, 10, -1, 0)]), 32, 32, 64, 64)
    with pytest.raises(native.NativeError, match='mode'):
        native.FusePlan(np.array([(0, 0, 10, 10, 0, 0)]), 32, 32, 64, 64, mode=7)
    with pytest.raises(native.NativeError, match='sizes'):
        native.FusePlan(np.array([(0, 0, 10, 10, 0, 0)]), 32, 32, 0, 64)


def test_empty_and_fully_clipped_plans():
    plan = native.FusePlan(np.zeros((0, 6), dtype=int), 8, 8, 16, 16)
    assert plan.n_spans == 1 and plan.covered_voxels == 0
    plan = native.FusePlan(np.array([(0, 0, 8, 8, 16, 16), (0, 0, 8, 8, 100, 3)]), 8, 8, 16, 16)
    assert plan.covered_voxels == 0
    np.testing.assert_array_equal(replay_overwrite(plan, np.ones((2, 8, 8), np.uint16)), np.zeros((16, 16), np.uint16))

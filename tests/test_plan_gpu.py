"""The work list of an overwrite plan produced on the device (csrc/plan_expand.hip: sq_fuse_plan_create_spans +
sq_fuse_plan_expand) is the host planner's table byte for byte, and fusing through it gives the same canvases."""
import numpy as np
import pytest
import torch

from image_stitcher_amd import native, placement
from test_plan_cpu import HDR, random_rects

pytestmark = pytest.mark.gpu


def both_tables(rects, th, tw, ch, cw):
    host = native.FusePlan(rects, th, tw, ch, cw)
    dev = native.FusePlan(rects, th, tw, ch, cw, expand_on_device=True)
    dev.device_table('cuda:0')
    return host, dev


def assert_same_table(host, dev):
    a, b = host.table, dev.table
    assert a.size == b.size == host.table_bytes == dev.table_bytes
    ha, hb = a[:HDR.itemsize].view(HDR)[0], b[:HDR.itemsize].view(HDR)[0]
    assert ha == hb, (ha, hb)
    if not np.array_equal(a, b):
        at = int(np.flatnonzero(a != b)[0])
        raise AssertionError(f'tables differ from byte {at} (items start at {ha["off_items"]}, seams at {ha["off_seams"]}, '
                             f'{ha["n_items"]} items)')


@pytest.mark.parametrize('seed', range(8))
def test_device_expansion_equals_host_plan_on_random_rectangles(seed):
    rng = np.random.default_rng(100 + seed)
    th, tw = (37, 53) if seed % 2 else (96, 300)
    ch, cw = int(rng.integers(40, 700)), int(rng.integers(40, 900))
    n = int(rng.integers(1, 40))
    host, dev = both_tables(random_rects(rng, n, th, tw, ch, cw), th, tw, ch, cw)
    assert_same_table(host, dev)


@pytest.mark.parametrize('g, shifts, missing', [
    (16, ((3, -244), (-244, -2)), ()),                  # the bench's config-3 geometry: 79 508 items
    (32, ((3, -244), (-244, -2)), ()),                  # the headline grid: 303 744 items, lanes of 258 row blocks
    (8, ((-5, -300), (-180, 7)), ((0, 0), (3, 4), (7, 7), (2, 2))),      # other signs, tiles missing
])
def test_device_expansion_equals_host_plan_on_registered_grids(g, shifts, missing):
    T = 2048
    sh = placement.Shifts(*shifts)
    wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=sh)
    rects = placement.grid_rects(g, g, T, T, sh)
    keep = [i for i in range(g * g) if (i // g, i % g) not in set(missing)]
    host, dev = both_tables(rects[keep], T, T, hc, wc)
    assert host.n_items > 10000
    assert_same_table(host, dev)


def test_wide_spans_are_cut_into_column_pieces_the_same_way():
    # spans wider than BLOCK_COLS (2048) and remainders narrower than a line: plan.cpp's column rule on the device
    rects = np.array([(0, 0, 40, 2100, 3, 5), (0, 0, 40, 2100, 20, 2090), (0, 0, 33, 2049, 50, 4100), (0, 0, 16, 4300, 90, 0)])
    host, dev = both_tables(rects, 64, 4300, 140, 6400)
    assert_same_table(host, dev)


def test_degenerate_plans():
    host, dev = both_tables(np.zeros((0, 6), dtype=int), 8, 8, 16, 16)            # nothing but zero fill
    assert_same_table(host, dev)
    host, dev = both_tables(np.array([(0, 0, 8, 8, 16, 16), (0, 0, 8, 8, 100, 3)]), 8, 8, 16, 16)   # every rect clipped away
    assert_same_table(host, dev)
    host, dev = both_tables(np.array([(0, 0, 8, 8, 0, 0)]), 8, 8, 8, 8)            # one tile = the canvas
    assert_same_table(host, dev)


def test_fusion_through_a_device_expanded_plan():
    rng = np.random.default_rng(3)
    g, T = 4, 256
    sh = placement.Shifts((2, -30), (-28, -3))
    wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=sh)
    rects = placement.grid_rects(g, g, T, T, sh)
    tiles = torch.from_numpy(rng.integers(0, 65536, size=(10, g * g, T, T), dtype=np.uint16)).cuda()
    flat = torch.from_numpy((0.5 + rng.random((T, T))).astype(np.float32)).cuda()
    outs = []
    for on_device in (False, True):
        plan = native.FusePlan(rects, T, T, hc, wc, expand_on_device=on_device)
        canvas = torch.full((10, hc, wc), 0xAAAA, dtype=torch.uint16, device='cuda')
        native.fuse_planes(plan, tiles, canvas, [flat] * 10)
        torch.cuda.synchronize()
        outs.append(canvas.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])


def test_unexpanded_plan_is_refused_by_the_fusion_entry_point():
    plan = native.FusePlan(np.array([(0, 0, 8, 8, 0, 0)]), 8, 8, 16, 16, expand_on_device=True)
    tiles = torch.zeros((1, 1, 8, 8), dtype=torch.uint16, device='cuda')
    canvas = torch.zeros((1, 16, 16), dtype=torch.uint16, device='cuda')
    # (device_table() would expand it: hand the entry point a buffer of the right size instead)
    import ctypes as C
    a = native._FuseArgs()
    C.memset(C.byref(a), 0, C.sizeof(a))
    dummy = torch.zeros(plan.table_bytes, dtype=torch.uint8, device='cuda')
    a.plan, a.table_dev, a.table_bytes = plan.handle, dummy.data_ptr(), plan.table_bytes
    a.canvas_dev, a.tile_base_dev = canvas.data_ptr(), tiles.data_ptr()
    assert native.lib().sq_fuse_planes(C.byref(a), native._stream_ptr()) != 0
    assert b'sq_fuse_plan_expand' in native.lib().sq_last_error()

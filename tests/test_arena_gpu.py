"""The canvas arena (sq_arena_create, csrc/arena.hip): device memory taken in physical slices, classified by a pair-fill probe and
mapped round-robin over the card's memory classes.  Whatever the mapping, it is plain device memory to every kernel: fusion into a
canvas carved from it gives the reference's voxels (golden canvases, oracle), tensors alias it, and it goes away cleanly."""
import os

import numpy as np
import pytest

from helpers import load_case, spec_of
from image_stitcher_amd import native, placement, synth
from oracle import stitch_oracle as O

pytestmark = pytest.mark.gpu
MiB = 1 << 20


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch


def test_arena_is_classified_balanced_and_aliases_like_plain_memory():
    torch = _torch()
    dev = torch.device('cuda:0')
    # 256 MiB chosen from up to 768 MiB of candidates: 8 MiB slices, 32 MiB probe units (the production sizes are 64 / 512 MiB)
    arena = native.DeviceArena(256 * MiB, dev, slice_bytes=8 * MiB, unit_bytes=32 * MiB, candidate_bytes=768 * MiB)
    info = arena.info
    # candidates are taken chunk by chunk until three classes hold a third of the arena each (after 2.5 x: two classes half each), 768 MiB at most
    assert info['bytes'] == 256 * MiB == arena.nbytes and info['n_slices'] == 32 and 32 <= info['n_candidates'] <= 96
    assert 1 <= info['n_classes'] <= native.SQ_ARENA_MAX_CLASSES
    assert sum(info['class_slices']) == 32 and sum(info['class_candidates']) == info['n_candidates']
    assert info['interleaved'] == (info['n_classes'] > 1) and info['probe_ms'] > 0 and info['max_pair_gbs'] >= info['min_pair_gbs'] > 0
    if info['n_classes'] > 1:      # round-robin: no class gives more than its share + 1 while another still has candidates
        fair = -(-32 // info['n_classes'])
        for have, took in zip(info['class_candidates'], info['class_slices']):
            assert took <= max(fair, 32 - sum(min(c, fair) for c in info['class_candidates']) + fair)
            assert took <= have
    assert arena.base % (2 * MiB) == 0
    # every byte of the range is writable and readable, across slice boundaries, and distinct (no slice mapped twice)
    whole = arena.take(arena.nbytes).view(torch.int64)
    assert whole.data_ptr() == arena.base and arena.free_bytes == 0
    whole.copy_(torch.arange(whole.numel(), dtype=torch.int64, device=dev))
    torch.cuda.synchronize()
    assert bool((whole[1:] - whole[:-1] == 1).all()) and int(whole[-1].item()) == whole.numel() - 1
    probe = whole[(8 * MiB // 8) - 2:(8 * MiB // 8) + 2].cpu().numpy()      # straddles the first slice boundary
    np.testing.assert_array_equal(probe, np.arange(8 * MiB // 8 - 2, 8 * MiB // 8 + 2))
    with pytest.raises(native.NativeError, match='left'):
        arena.take(1)
    assert arena.in_use()
    del whole
    assert not arena.in_use()
    arena.reset()
    # bump allocation: aligned, disjoint, views keep the range "in use"
    a = arena.take(1000)
    b = arena.take(24, align=4096)
    assert a.data_ptr() == arena.base and b.data_ptr() == arena.base + 4096 and a.dtype == torch.uint8
    v = b.view(torch.uint16)[3:7]
    del a, b
    assert arena.in_use()
    del v
    assert not arena.in_use()
    arena.close()
    arena.close()      # idempotent
    with pytest.raises(native.NativeError, match='closed'):
        arena.take(1)
    # the control: slices in creation order, no probe
    nat = native.DeviceArena(64 * MiB, dev, slice_bytes=8 * MiB, unit_bytes=32 * MiB, natural_order=True)
    assert nat.info['n_classes'] == 1 and not nat.info['interleaved'] and nat.info['n_candidates'] == 8 and nat.info['probe_ms'] == 0
    nat.close()
    with pytest.raises(native.NativeError, match='multiple of 2 MiB'):
        native.DeviceArena(64 * MiB, dev, slice_bytes=3 * MiB)
    with pytest.raises(native.NativeError, match='positive'):
        native.DeviceArena(0, dev)


def test_stitcher_uses_the_arena_and_reuses_it(tmp_path, monkeypatch):
    """Stitcher.stitch_region with the arena forced on for a small golden acquisition: the reference's canvas, the arena kept for
    the next region, left alone while a caller still holds the last device canvas, and a plain allocation below the threshold."""
    torch = _torch()
    from image_stitcher_amd.stitcher import Stitcher
    from image_stitcher_amd.stitcher_parameters import StitchingParameters
    info, arrays = load_case('reg_multi')
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec_of(info), root)
    monkeypatch.setenv('SQ_CANVAS_ARENA_MIN_BYTES', '1')
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization=None)
    assert st.canvas_arena_min_bytes == 1
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    st.calculate_shifts(st.timepoints[0], st.regions[0])
    keys = list(info['canvases'])
    first = None
    for key in keys:
        t, region = key[1:].split('_', 1)
        got = st.stitch_region(int(t), region)
        np.testing.assert_array_equal(got, arrays[f'{key}_canvas'])
        assert st._arena is not None and not st._arena.in_use()      # the host copy left: the arena is free for the next region
        first = first or st._arena
        assert st._arena is first                                     # ... and it IS reused
    assert st.canvas_arena_info['n_slices'] >= 1
    # a caller that keeps the device canvas: the next region must not overwrite it
    t, region = keys[0][1:].split('_', 1)
    held = st.stitch_region(int(t), region, device_output=True)
    assert st._arena.in_use()
    t2, region2 = keys[1][1:].split('_', 1)
    other = st.stitch_region(int(t2), region2)
    assert st._arena is not first
    np.testing.assert_array_equal(other, arrays[f'{keys[1]}_canvas'])
    np.testing.assert_array_equal(held.cpu().numpy(), arrays[f'{keys[0]}_canvas'])
    del held
    # below the threshold: a plain allocation, no arena
    plain = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization=None)
    plain.canvas_arena_min_bytes = 1 << 40
    plain.get_timepoints(); plain.extract_acquisition_parameters(); plain.get_pixel_size(); plain.parse_acquisition_metadata()
    plain.calculate_shifts(plain.timepoints[0], plain.regions[0])
    np.testing.assert_array_equal(plain.stitch_region(int(t), region), arrays[f'{keys[0]}_canvas'])
    assert plain._arena is None


@pytest.mark.parametrize('seed', range(3))
def test_plane_groups_into_an_arena_canvas_equal_the_oracle(seed):
    """Random geometry, several planes that share gains (plane groups + seam owners + device queues), canvas carved from an
    arena at an odd offset: every voxel the oracle's, the padding between the planes untouched."""
    torch = _torch()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(100 + seed)
    spec = synth.GridSpec(rows=3, cols=4, tile_h=96, tile_w=160, ov_y=20, ov_x=36, seed=50 + seed)
    shifts = placement.Shifts((int(rng.integers(-3, 4)), -36), (-20, int(rng.integers(-3, 4))))
    rects = placement.grid_rects(3, 4, 160, 96, shifts)
    wc, hc = placement.canvas_size(3, 4, 160, 96, use_registration=True, shifts=shifts)
    n_planes = 7
    tiles_np = np.stack([np.roll(spec.tile_stack(), 3 * p, axis=2) for p in range(n_planes)])
    flat_np = synth.synthetic_flatfield(96, 160, np.float32)
    arena = native.DeviceArena(64 * MiB, dev, slice_bytes=8 * MiB, unit_bytes=32 * MiB, candidate_bytes=128 * MiB)
    arena.take(12345)                                          # the canvas does not start at the arena's base
    guard = arena.take(arena.free_bytes - 4096, align=512)
    guard.fill_(0xA5)
    arena.reset()
    arena.take(12345)
    canvas = native.empty_canvas(n_planes, hc, wc, torch.uint16, dev, arena=arena)
    assert canvas.data_ptr() % 128 == 0 and canvas.data_ptr() > arena.base
    plan = native.FusePlan(rects, 96, 160, hc, wc)
    flat = torch.from_numpy(flat_np).to(dev)
    tiles = torch.from_numpy(tiles_np).to(dev)
    native.fuse_planes(plan, tiles, canvas, [flat] * n_planes, flags=native.SQ_FUSE_FORCE_QUEUES)
    torch.cuda.synchronize()
    for p in range(n_planes):
        want = O.fuse_plane_overwrite(list(tiles_np[p]), rects, hc, wc, flat_np)
        np.testing.assert_array_equal(canvas[p].cpu().numpy(), want)
    # the bytes between the planes (stride padding) and after the last plane still hold the guard pattern
    stride_b = canvas.stride(0) * 2
    raw = torch.as_tensor(native._RawDeviceMemory(arena, canvas.data_ptr(), stride_b * n_planes + 1024), device=dev)
    for p in range(n_planes):
        pad = raw[p * stride_b + hc * wc * 2:(p + 1) * stride_b]
        assert bool((pad == 0xA5).all())
    assert bool((raw[stride_b * n_planes:] == 0xA5).all())
    del raw, canvas, guard
    arena.close()


def test_streamed_run_fuses_into_arena_canvases(tmp_path, monkeypatch):
    """Stitcher.run() with .ome.zarr output (the CLI's default path) fuses into the stream writer's two slot canvases; with the
    arena forced on they are carved from a DeviceArena too.  The stores are the reference's canvases, and a card that cannot give
    the slices falls back to a plain allocation instead of failing."""
    torch = _torch()
    from image_stitcher_amd import omezarr
    from image_stitcher_amd.stitcher import Stitcher
    from image_stitcher_amd.stitcher_parameters import StitchingParameters
    info, arrays = load_case('reg_multi')
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec_of(info), root)
    monkeypatch.setenv('SQ_CANVAS_ARENA_MIN_BYTES', '1')
    made = []
    real = native.DeviceArena

    class Counting(real):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made.append(self)
    monkeypatch.setattr(native, 'DeviceArena', Counting)
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization=None)
    st.run()
    assert len(made) == 1, "one arena for the writer's slots, reused for every region of the same geometry"
    for key in info['canvases']:
        t, region = key[1:].split('_', 1)
        store = os.path.join(st.output_folder, f'{t}_stitched', f'{region}_stitched.ome.zarr')
        np.testing.assert_array_equal(omezarr.read_array(os.path.join(store, '0')), arrays[f'{key}_canvas'])
    # the card "cannot give the slices": a plain allocation, the same stores
    def refusing(*a, **k):
        raise native.NativeError("sq_arena_create(1 bytes) failed: sq_arena_create: hipMemCreate of slice 0 of 1 (64 MiB each) failed: out of memory")
    monkeypatch.setattr(native, 'DeviceArena', refusing)
    st2 = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization=None)
    st2.run()
    key = list(info['canvases'])[0]
    t, region = key[1:].split('_', 1)
    store = os.path.join(st2.output_folder, f'{t}_stitched', f'{region}_stitched.ome.zarr')
    np.testing.assert_array_equal(omezarr.read_array(os.path.join(store, '0')), arrays[f'{key}_canvas'])
    assert not st2._arena_unsupported       # out of memory is not "unsupported": the next region may try again
    # ... while a platform without virtual memory management is remembered
    def unsupported(*a, **k):
        raise native.NativeError("sq_arena_create failed: sq_arena_create: virtual memory management unsupported: hipMemCreate ...")
    monkeypatch.setattr(native, 'DeviceArena', unsupported)
    st3 = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization=None)
    st3.get_timepoints(); st3.extract_acquisition_parameters(); st3.get_pixel_size(); st3.parse_acquisition_metadata()
    st3.calculate_shifts(st3.timepoints[0], st3.regions[0])
    np.testing.assert_array_equal(st3.stitch_region(int(t), region), arrays[f'{key}_canvas'])
    assert st3._arena_unsupported and st3._arena is None


def test_an_address_that_was_mapped_is_never_mapped_again():
    """Arenas created and destroyed one after the other: no arena comes back inside an address range an earlier one (or its
    candidates) had mapped -- such ranges are retired, not freed -- and a large device-to-host copy of what a kernel has just
    written into the new arena reads exactly that (with freed-and-reused ranges the copy engines read the slices that USED to be
    mapped there: a freshly fused canvas came back as an earlier arena's zeros)."""
    torch = _torch()
    dev = torch.device('cuda:0')
    seen = []
    for k in range(6):
        arena = native.DeviceArena(48 * MiB, dev, slice_bytes=8 * MiB, unit_bytes=16 * MiB, candidate_bytes=96 * MiB)
        lo, hi = arena.base, arena.base + arena.nbytes
        for a, b in seen:
            assert hi <= a or lo >= b, f'arena {k} at [{lo:#x}, {hi:#x}) overlaps the retired range [{a:#x}, {b:#x})'
        seen.append((lo, hi))
        t = arena.take(40 * MiB).view(torch.int32)
        t.fill_(0x01010101 * (k + 1))                      # a kernel writes through the page tables
        torch.cuda.synchronize()
        host = t.cpu().numpy()                              # the runtime's copy of 40 MiB, across slice boundaries
        assert (host == 0x01010101 * (k + 1)).all(), f'arena {k}: the copy read something else than the kernel wrote'
        del t
        arena.close()


def test_two_classes_flag_maps_the_two_largest_classes_only():
    """SQ_ARENA_TWO_CLASSES (the measurement aid behind the arena's stop rule -- two halves write as fast as three thirds,
    profiles/r04_exp_arena_two_classes.log): when the two largest classes can fill the arena, no slice of another class is mapped;
    the memory is as usable as any arena's."""
    torch = _torch()
    dev = torch.device('cuda:0')
    arena = native.DeviceArena(256 * MiB, dev, slice_bytes=8 * MiB, unit_bytes=32 * MiB, candidate_bytes=768 * MiB, two_classes=True)
    info = arena.info
    assert sum(info['class_slices']) == 32
    by_size = sorted(range(len(info['class_candidates'])), key=lambda c: -info['class_candidates'][c])
    if len(by_size) > 2 and info['class_candidates'][by_size[0]] + info['class_candidates'][by_size[1]] >= 32:
        assert all(info['class_slices'][c] == 0 for c in by_size[2:]), info
    t = arena.take(arena.nbytes).view(torch.int32)
    t.fill_(0x1234567)
    torch.cuda.synchronize()
    assert bool((t.cpu() == 0x1234567).all())
    del t
    arena.close()

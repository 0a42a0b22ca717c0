"""BASELINE.json's configurations as parity cases (scaled where the full size would not finish in
seconds on the oracle), plus size-independent properties at the full tile size."""
import os

import numpy as np
import pytest

from image_stitcher_amd import native, omezarr, placement, registration, sharding, synth
from image_stitcher_amd.stitcher import Stitcher
from image_stitcher_amd.stitcher_parameters import StitchingParameters
from image_stitcher_amd.tiffio import read_image
from oracle import stitch_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _device_tiles(spec, z=0, ch=0, region_idx=0, t=0):
    g = spec.rows * spec.cols
    desc = np.zeros(g, dtype=native.SYNTH_DTYPE)
    for r in range(spec.rows):
        for c in range(spec.cols):
            oy, ox = spec.origin(r, c)
            desc[r * spec.cols + c] = (spec.scene_seed(region_idx, t, z, ch) % 2 ** 64,
                                       spec.noise_seed(region_idx, t, z, ch, spec.fov_index(r, c)) % 2 ** 64, oy, ox)
    return native.synth_tiles(desc, spec.tile_h, spec.tile_w, spec.noise, spec.dtype, _dev())


def test_config4_geometry_32x32_grid_two_planes():
    """32x32 grid (1024 tiles, the planner's largest configuration) at 1/16 tile size, planes
    dealt to two 'ranks' block-cyclically: the union equals the oracle's canvases."""
    import torch
    spec = synth.GridSpec(rows=32, cols=32, tile_h=128, tile_w=128, ov_y=16, ov_x=16, seed=4000)
    shifts = placement.Shifts((3, -16), (-16, -2))
    order = placement.filename_order([spec.fov_index(r, c) for r in range(32) for c in range(32)])
    order_rc = [divmod(i, 32) for i in order]
    rects = placement.grid_rects(32, 32, 128, 128, shifts, order=order_rc)
    wc, hc = placement.canvas_size(32, 32, 128, 128, use_registration=True, shifts=shifts)
    plan = native.FusePlan(rects, 128, 128, hc, wc)
    n_planes = 3
    idx = torch.tensor(order, device=_dev())
    # torch has no uint16 gather: reorder through an int16 view
    tiles = torch.stack([_device_tiles(spec, z=p).view(torch.int16)[idx] for p in range(n_planes)]).view(torch.uint16)
    canvas = torch.zeros((n_planes, hc, wc), dtype=torch.uint16, device=_dev())
    for rank in range(2):
        for p in sharding.block_cyclic(n_planes, rank, 2):
            native.fuse_planes(plan, tiles[p:p + 1].contiguous(), canvas[p:p + 1])
    torch.cuda.synchronize()
    got = canvas.cpu().numpy()
    for p in range(n_planes):
        host = tiles[p].cpu().numpy()
        np.testing.assert_array_equal(got[p], O.fuse_plane_overwrite(list(host), rects, hc, wc))


def test_config2_full_size_properties():
    """8x8 grid of 2048^2 tiles with -r (config 2) at full size: registration of every pair agrees
    with the planted drift, and fusion satisfies size-independent properties -- every tile's
    interior appears verbatim at its placement, uncovered rows are zero, and a second launch into
    a poisoned canvas gives the identical result (every voxel is written exactly once)."""
    import torch
    spec = synth.GridSpec(rows=8, cols=8, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=2000)
    tiles = _device_tiles(spec)
    xs = [spec.stage_mm(0, c)[0] for c in range(8)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(8)]
    shifts = registration.register_grid_center(tiles, 8, 8, xs, ys, spec.pixel_size_um, spec.pixel_binning, 'phase')
    assert (shifts.h_shift, shifts.v_shift) == ((3, -244), (-244, -2))
    mx, my = placement.registration_crop_widths(xs, ys, 2048, 2048, spec.pixel_size_um, spec.pixel_binning)
    (hp, (h0, h1)), (vp, (v0, v1)) = registration.all_pairs(8, 8, 2048, 2048, mx, my)
    hs, _, _ = registration.register_pairs(tiles, hp, h0, h1, 10, 'phase')
    vs, _, _ = registration.register_pairs(tiles, vp, v0, v1, 10, 'phase')
    truth_h = np.array([3.0, 256 - 244.0]); truth_v = np.array([256 - 244.0, -2.0])
    assert np.sqrt(((hs - truth_h) ** 2).mean()) <= 0.5 and np.sqrt(((vs - truth_v) ** 2).mean()) <= 0.5   # shift RMSE bar
    rects = placement.grid_rects(8, 8, 2048, 2048, shifts)
    wc, hc = placement.canvas_size(8, 8, 2048, 2048, use_registration=True, shifts=shifts)
    plan = native.FusePlan(rects, 2048, 2048, hc, wc)
    canvas = torch.full((1, hc, wc), 0xAAAA, dtype=torch.uint16, device=_dev())
    native.fuse_planes(plan, tiles[None], canvas)
    first = canvas.clone()
    canvas.fill_(0x5555)
    native.fuse_planes(plan, tiles[None], canvas)
    assert torch.equal(first.view(torch.int16), canvas.view(torch.int16))
    # voxel census: covered voxels are >= 800 (the generator's floor), the rest are exactly zero
    nz = int((canvas.view(torch.int16) != 0).sum())
    assert nz == plan.covered_voxels
    for i in (0, 9, 27, 63):
        sy, sx, h, w, dy, dx = rects[i]
        np.testing.assert_array_equal(canvas[0, dy + 300:dy + 364, dx + 300:dx + 364].cpu().numpy(),
                                      tiles[i, sy + 300:sy + 364, sx + 300:sx + 364].cpu().numpy())


def test_config5_hcs_plate_scaled(tmp_path):
    """HCS plate (config 5) scaled down: wells x timepoints, 5x5 tiles, 3 channels; per-well
    registration (the north-star extension) and fusion against the oracle for every (t, well)."""
    wells = ('A1', 'A2', 'B1')
    spec = synth.GridSpec(rows=5, cols=5, tile_h=96, tile_w=128, ov_y=24, ov_x=32, seed=5000, regions=wells, nt=2,
                          channels=synth.DEFAULT_CHANNELS[:3])
    root = str(tmp_path / 'plate')
    synth.write_acquisition(spec, root)
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization='phase')
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    acq = O.parse_acquisition(root, read_image)
    assert st.regions == list(wells) == acq.regions and st.num_t == 2
    table = []
    for t in st.timepoints:
        for well in st.regions:
            st.calculate_shifts(t, well)                       # per-well registration
            want = O.calculate_shifts(acq, t, well, read_image, '', 0, 'Unidirectional', 'phase')
            assert (st.h_shift, st.v_shift) == (want['h_shift'], want['v_shift'])
            table.append(sharding.shifts_to_row(st._shifts()))
            canvas = st.stitch_region(int(t), well)
            np.testing.assert_array_equal(canvas, O.stitch_region(acq, t, well, read_image, True, want))
    assert np.stack(table).shape == (6, sharding.SHIFT_ROW)


def test_config3_planes_from_files_to_store_at_full_size(tmp_path):
    """BASELINE config 3 end to end at full tile and grid size, cut to one channel x two z planes: 512
    files of 2048^2 -> CLI (-r -ff semantics with supplied gains) -> streamed multiscale OME-Zarr store.
    Level 0 equals the oracle's canvas plane by plane (the oracle itself reproduces the genuine reference
    on exactly this configuration), the pyramid levels are the strided gathers of it."""
    import torch
    dev = torch.device('cuda:0')
    spec = synth.GridSpec(rows=16, cols=16, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=3000, nz=2)
    root = str(tmp_path / 'acq')
    synth.write_acquisition_device(spec, root, dev)
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True, apply_flatfield=True), normalization='phase',
                  zarr_compression='none')
    st.output_folder = str(tmp_path / 'out')
    st.batch_bytes_limit = 256 * 2048 * 2048 * 2          # one plane per batch: both writer slots are used
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    flat = synth.synthetic_flatfield(2048, 2048, np.float32)
    st.flatfields = {0: flat}
    st.calculate_shifts(0, 'R0')
    assert (st.h_shift, st.v_shift) == ((3, -244), (-244, -2))
    path = st.stream_region_to_zarr(0, 'R0')
    assert st.num_pyramid_levels == 6
    acq = O.parse_acquisition(root, read_image)
    shifts = {'h_shift': st.h_shift, 'v_shift': st.v_shift}
    want = O.stitch_region(acq, 0, 'R0', read_image, True, shifts, {0: flat}, True)
    assert want.shape == (1, 1, 2, 36473, 29138)
    got = omezarr.read_array(os.path.join(path, '0'))
    np.testing.assert_array_equal(got, want)
    for lv, lvl in enumerate(O.pyramid_nearest(want, 6)):
        if lv:
            np.testing.assert_array_equal(omezarr.read_array(os.path.join(path, str(lv))), lvl)


def test_stitch_region_large_planes_distinct_gains_per_channel(tmp_path):
    """planes big enough that a fusion launch is still running when the next one's pointer
    table is uploaded.  2 channels x 2 z of a 4x4 grid of 2048^2 tiles, a different gain image per channel,
    tiny ingest batches (one plane each) so launches follow one another back to back through stitch_region;
    every plane must come out divided by ITS channel's gains."""
    import torch
    dev = torch.device('cuda:0')
    spec = synth.GridSpec(rows=4, cols=4, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=4100, nz=2,
                          channels=synth.DEFAULT_CHANNELS[:2])
    root = str(tmp_path / 'acq')
    synth.write_acquisition_device(spec, root, dev)
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True, apply_flatfield=True), normalization='phase')
    st.batch_bytes_limit = 16 * 2048 * 2048 * 2          # one plane per batch -> four launches in a row
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    base = synth.synthetic_flatfield(2048, 2048, np.float32)
    flats = {0: base, 1: (base[::-1, ::-1] * np.float32(1.75)).copy()}
    st.flatfields = flats
    st.calculate_shifts(0, 'R0')
    assert (st.h_shift, st.v_shift) == ((3, -244), (-244, -2))
    got = st.stitch_region(0, 'R0')
    acq = O.parse_acquisition(root, read_image)
    want = O.stitch_region(acq, 0, 'R0', read_image, True, {'h_shift': st.h_shift, 'v_shift': st.v_shift}, flats, True)
    assert got.shape == want.shape == (1, 2, 2, want.shape[3], want.shape[4])
    for c in range(2):
        for z in range(2):
            np.testing.assert_array_equal(got[0, c, z], want[0, c, z], err_msg=f'channel {c} z {z}')


def test_config4_full_size_plane_and_25_planes_through_the_work_queues():
    """BASELINE config 4 at FULL size: one (c, z) plane of the 32x32 grid of 2048^2 tiles (1024 tiles -> a
    73193 x 58034 canvas, a 303 000-item plan, float32 gains, shifts registered on the device) equals the oracle
    voxel for voxel -- and the same plan at 25 planes (what one of 8 GPUs fuses: 7.6 M work items through the 32-bit
    queue arithmetic, plane groups of 5 + partial groups) into a poisoned canvas: every plane is the verified one
    for its gain image, nothing outside the planes is touched.  The 25 planes read the same tile stack (a pointer
    table with repeated entries), so the oracle runs twice, not 25 times."""
    import torch
    dev = _dev()
    torch.cuda.empty_cache()
    g, T = 32, 2048
    spec = synth.GridSpec(rows=g, cols=g, tile_h=T, tile_w=T, ov_y=244, ov_x=244, seed=4000)
    tiles = _device_tiles(spec)                                       # [1024, 2048, 2048], storage order
    xs = [spec.stage_mm(0, c)[0] for c in range(g)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(g)]
    shifts = registration.register_grid_center(tiles, g, g, xs, ys, spec.pixel_size_um, spec.pixel_binning, 'phase')
    assert (shifts.h_shift, shifts.v_shift) == ((3, -244), (-244, -2))
    order = placement.filename_order([spec.fov_index(r, c) for r in range(g) for c in range(g)])
    order_rc = [divmod(i, g) for i in order]
    rects = placement.grid_rects(g, g, T, T, shifts, order=order_rc)
    wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=shifts)
    # the oracle's own geometry (nothing of placement.py): canvas size and rectangles
    ow, oh, _ = O.output_dimensions(list(xs), list(ys), T, T, spec.pixel_size_um, True, shifts.h_shift, shifts.v_shift, None, 1)
    assert (oh, ow) == (hc, wc) == (73193, 58034)
    want_rects = []
    for i in order:
        r, c = divmod(i, g)
        x_px, y_px, top, bottom, left, right = O.tile_rect({'x': xs[c], 'y': ys[r]}, list(xs), list(ys), T, T, spec.pixel_size_um,
                                                           True, shifts.h_shift, shifts.v_shift, None, 0, wc, hc)
        want_rects.append((top, left, T - top - bottom, T - left - right, y_px + top, x_px + left))
    want_rects = np.array(want_rects, dtype=np.int64)
    np.testing.assert_array_equal(rects, want_rects)
    plan = native.FusePlan(rects, T, T, hc, wc)
    assert plan.n_items > 300000
    n_planes = 25
    base = synth.synthetic_flatfield(T, T, np.float32)
    gains = [base, (base * np.float32(1.03125)).astype(np.float32)]
    d_gains = [torch.from_numpy(f).to(dev) for f in gains]
    which = [0] * 13 + [1] * 12
    canvas = native.empty_canvas(n_planes, hc, wc, torch.uint16, dev)
    storage = torch.as_strided(canvas, (n_planes * canvas.stride(0),), (1,))
    storage.view(torch.int16).fill_(-16657)                           # 0xBEEF everywhere, the gaps between planes too
    esz = T * T * 2
    ptrs_plane = tiles.data_ptr() + torch.tensor(order, dtype=torch.int64) * esz
    ptrs = ptrs_plane.repeat(n_planes).to(dev)                        # every plane reads the same 1024 tiles
    assert n_planes * plan.n_items > 7_000_000
    native.fuse_planes(plan, None, canvas, [d_gains[k] for k in which], tile_ptrs=ptrs)
    torch.cuda.synchronize()
    host = tiles.cpu().numpy()
    for k, first in ((0, 0), (1, 13)):
        want = O.fuse_plane_overwrite([host[i] for i in order], want_rects, hc, wc, gains[k])
        got = canvas[first].cpu().numpy()
        assert np.array_equal(got, want), f'plane {first} (gain image {k}) differs from the oracle'
        assert int(np.count_nonzero(want)) <= plan.covered_voxels
        del want, got
    ref = {0: canvas[0].view(torch.int16), 1: canvas[13].view(torch.int16)}
    for p in range(n_planes):
        assert torch.equal(canvas[p].view(torch.int16), ref[which[p]]), f'plane {p} differs from its verified twin'
    gap = canvas.stride(0) - hc * wc
    if gap:
        pads = torch.as_strided(canvas, (n_planes, gap), (canvas.stride(0), 1), storage_offset=canvas.storage_offset() + hc * wc)
        assert bool((pads.view(torch.int16) == -16657).all()), "the padding between planes was written"
    # one plane alone (no plane groups: the per-plane kernel with its queues) gives the same voxels
    single = native.empty_canvas(1, hc, wc, torch.uint16, dev)
    native.fuse_planes(plan, None, single, [d_gains[0]], tile_ptrs=ptrs[:g * g])
    assert torch.equal(single[0].view(torch.int16), ref[0])
    del canvas, single, tiles
    torch.cuda.empty_cache()


def test_config5_full_plate_96_wells_per_well_registration(tmp_path):
    """BASELINE config 5 in its full SHAPE: a 96-well plate (A1..H12), 5x5 tiles per well, 3 channels, per-well
    registration, streamed to OME-Zarr -- with small tiles (96 x 128) and T = 2 so that the 14 400 files are written
    in seconds (the voxel count scales with the tile size, the bookkeeping does not: 192 (timepoint, well) units,
    192 registrations, 192 stores, one 192-row shift table).  A sample of wells is compared with the oracle voxel for
    voxel (store read through the independent spec-level reader); every unit has its row in shift_table.json."""
    import json
    import blosc_ref
    wells = tuple(f'{r}{c}' for r in 'ABCDEFGH' for c in range(1, 13))
    spec = synth.GridSpec(rows=5, cols=5, tile_h=96, tile_w=128, ov_y=24, ov_x=32, seed=5100, regions=wells, nt=2,
                          channels=synth.DEFAULT_CHANNELS[:3])
    root = str(tmp_path / 'plate')
    synth.write_acquisition(spec, root)
    from image_stitcher_amd import stitcher_cli
    stitcher_cli.main(['-i', root, '-r', '--per-region-registration'])
    (out,) = [d for d in os.listdir(tmp_path) if d.startswith('plate_stitched_')]
    with open(os.path.join(tmp_path, out, 'shift_table.json')) as fh:
        table = json.load(fh)
    assert table['per_region_registration'] is True and len(table['shifts']) == 192
    assert [(e['timepoint'], e['region']) for e in table['shifts']] == [(t, w) for t in (0, 1) for w in sorted(wells)]
    acq = O.parse_acquisition(root, read_image)
    rows = {(e['timepoint'], e['region']): e for e in table['shifts']}
    for t, well in ((0, 'A1'), (0, 'D7'), (1, 'H12'), (1, 'B10')):
        want_s = O.calculate_shifts(acq, t, well, read_image, '', 0, 'Unidirectional', 'phase')
        e = rows[(t, well)]
        assert (tuple(e['h_shift']), tuple(e['v_shift'])) == (tuple(want_s['h_shift']), tuple(want_s['v_shift']))
        want = O.stitch_region(acq, t, well, read_image, True, want_s)
        store = os.path.join(tmp_path, out, f'{t}_stitched', f'{well}_stitched.ome.zarr')
        got, meta, _ = blosc_ref.read_zarr_v2_array(os.path.join(store, '0'))
        np.testing.assert_array_equal(got, want)
        assert meta['compressor']['id'] == 'blosc'
    stores = sum(1 for t in (0, 1) for w in wells
                 if os.path.isfile(os.path.join(tmp_path, out, f'{t}_stitched', f'{w}_stitched.ome.zarr', '.zattrs')))
    assert stores == 192


def test_config5_one_well_at_real_tile_size(tmp_path):
    """BASELINE config 5 at its REAL tile size: two wells of 5 x 5 tiles of 2048 x 2048 pixels x 3
    channels, T = 2 (300 files, 2.5 GB, written from the device generator), through the CLI with ``-r
    --per-region-registration`` exactly like the plate run.  Every (timepoint, well) unit's shifts and level-0 voxels equal
    the oracle's (the store read through the independent spec-level reader); the pyramid of one unit equals the oracle's."""
    import json
    import torch
    import blosc_ref
    spec = synth.GridSpec(rows=5, cols=5, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, jy=3, jx=-2, seed=5200, regions=('C4', 'F11'),
                          nt=2, channels=synth.DEFAULT_CHANNELS[:3])
    root = str(tmp_path / 'wells')
    paths = synth.write_acquisition_device(spec, root, torch.device('cuda:0'))
    assert len(paths) == 2 * 2 * 25 * 3
    from image_stitcher_amd import stitcher_cli
    stitcher_cli.main(['-i', root, '-r', '--per-region-registration'])
    (out,) = [d for d in os.listdir(tmp_path) if d.startswith('wells_stitched_')]
    with open(os.path.join(tmp_path, out, 'shift_table.json')) as fh:
        table = json.load(fh)
    assert table['per_region_registration'] is True
    assert [(e['timepoint'], e['region']) for e in table['shifts']] == [(0, 'C4'), (0, 'F11'), (1, 'C4'), (1, 'F11')]
    acq = O.parse_acquisition(root, read_image)
    for e in table['shifts']:
        t, well = e['timepoint'], e['region']
        want_s = O.calculate_shifts(acq, t, well, read_image, '', 0, 'Unidirectional', 'phase')
        assert (tuple(e['h_shift']), tuple(e['v_shift'])) == (tuple(want_s['h_shift']), tuple(want_s['v_shift'])) == ((3, -244), (-244, -2))
        want = O.stitch_region(acq, t, well, read_image, True, want_s)
        assert want.shape[1:3] == (3, 1) and want.shape[3] > 9000 and want.shape[4] > 9000
        store = os.path.join(tmp_path, out, f'{t}_stitched', f'{well}_stitched.ome.zarr')
        got, meta, _ = blosc_ref.read_zarr_v2_array(os.path.join(store, '0'))
        np.testing.assert_array_equal(got, want)
        assert meta['compressor']['id'] == 'blosc'
        if (t, well) == (1, 'F11'):
            levels = O.pyramid_nearest(want, len([d for d in os.listdir(store) if d.isdigit()]))
            assert len(levels) >= 4
            for lv, wl in enumerate(levels):
                np.testing.assert_array_equal(blosc_ref.read_zarr_v2_array(os.path.join(store, str(lv)))[0], wl)
        del want, got

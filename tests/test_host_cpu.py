"""Host logic that needs no GPU: parameters, CLI flag surface, placement geometry against the
oracle and the golden placements, metadata parsing, TIFF and OME-Zarr round trips."""
import json
import os

import numpy as np
import pytest

from helpers import REGION_CASES, load_case, spec_of
from image_stitcher_amd import ometiff, omezarr, placement, synth, tiffio
from image_stitcher_amd.stitcher_parameters import StitchingParameters
from image_stitcher_amd import stitcher_cli
from oracle import stitch_oracle as O


def test_parameters_validation_and_json(tmp_path):
    p = StitchingParameters(input_folder=str(tmp_path), use_registration=True, registration_channel=None)
    p.validate()
    assert p.registration_channel == ''
    assert p.stitched_folder.startswith(str(tmp_path) + "_stitched_")
    j = tmp_path / 'p.json'
    p.to_json(str(j))
    q = StitchingParameters.from_json(str(j))
    assert q.to_dict() == p.to_dict()
    assert StitchingParameters.from_dict({'input_folder': str(tmp_path), 'bogus': 1}).input_folder == str(tmp_path)
    with pytest.raises(ValueError, match='does not exist'):
        StitchingParameters(input_folder=str(tmp_path / 'nope')).validate()
    with pytest.raises(ValueError, match='Output format'):
        StitchingParameters(input_folder=str(tmp_path), output_format='.png').validate()
    with pytest.raises(ValueError, match='Scan pattern'):
        StitchingParameters(input_folder=str(tmp_path), scan_pattern='Zigzag').validate()
    with pytest.raises(ValueError, match='non-negative'):
        StitchingParameters(input_folder=str(tmp_path), use_registration=True, registration_z_level=-1).validate()


def test_cli_flags_match_reference(tmp_path):
    a = stitcher_cli.parse_args(['-i', str(tmp_path), '-r', '-ff', '--registration-channel', '488',
                                 '--registration-z-level', '2', '-s', 'S-Pattern', '-mt', '-mw', '-f', '.ome.zarr',
                                 '--dynamic-registration'])
    p = stitcher_cli.create_params(a)
    assert (p.use_registration, p.apply_flatfield, p.registration_channel, p.registration_z_level) == (True, True, '488', 2)
    assert (p.scan_pattern, p.merge_timepoints, p.merge_hcs_regions, p.dynamic_registration) == ('S-Pattern', True, True, True)
    with pytest.raises(SystemExit):
        stitcher_cli.parse_args([])   # -i is required


def test_tiff_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    for arr in (rng.integers(0, 65536, (33, 47)).astype(np.uint16), rng.integers(0, 256, (20, 31)).astype(np.uint8),
                rng.integers(0, 256, (12, 9, 3)).astype(np.uint8)):
        p = str(tmp_path / 'a.tiff')
        tiffio.write_tiff(p, arr)
        np.testing.assert_array_equal(tiffio.read_image(p), arr)
    with pytest.raises(ValueError):
        tiffio.write_tiff(str(tmp_path / 'b.tiff'), np.zeros((4, 4), np.float32))


def _strip_tiff(arr, rows_per_strip, big_endian=False, ifd_last=True, gap=0):
    """An uncompressed TIFF the way libtiff lays it out: strips of a few rows first, the IFD (with its offset / count
    arrays) after the pixel data; ``gap`` bytes between two strips make them non-contiguous."""
    import struct
    bo = '>' if big_endian else '<'
    h, w = arr.shape
    data = arr.astype(arr.dtype.newbyteorder(bo)).tobytes()
    row = w * arr.dtype.itemsize
    strips = [data[r * row:(r + rows_per_strip) * row] for r in range(0, h, rows_per_strip)]
    body, offs = b'', []
    pos = 8
    for i, st in enumerate(strips):
        if i == 1 and gap:
            body += b'\0' * gap
            pos += gap
        offs.append(pos)
        body += st
        pos += len(st)
    n = len(strips)
    ifd = pos
    arrays_at = ifd + 2 + 9 * 12 + 4
    ent = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, arr.dtype.itemsize * 8), (259, 3, 1, 1), (262, 3, 1, 1),
           (273, 4, n, arrays_at if n > 1 else offs[0]), (277, 3, 1, 1), (278, 4, 1, rows_per_strip),
           (279, 4, n, arrays_at + 4 * n if n > 1 else len(strips[0]))]
    out = (b'MM' if big_endian else b'II') + struct.pack(bo + 'HI', 42, ifd) + body + struct.pack(bo + 'H', len(ent))
    for tag, typ, cnt, val in ent:
        out += struct.pack(bo + 'HHI', tag, typ, cnt) + (struct.pack(bo + 'HH', val, 0) if typ == 3 else struct.pack(bo + 'I', val))
    out += struct.pack(bo + 'I', 0)
    if n > 1:
        out += struct.pack(bo + f'{n}I', *offs) + struct.pack(bo + f'{n}I', *[len(st) for st in strips])
    return out


def test_tiff_read_into_staging(tmp_path):
    """``read_image_into`` puts a tile's pixels straight into the caller's buffer (the ingest's page-locked staging) for
    the files Squid writes -- one strip or many contiguous ones, IFD before or after the data -- and declines, leaving
    the buffer alone, for everything else (then ``read_image`` decodes it as before)."""
    rng = np.random.default_rng(3)
    a16 = rng.integers(0, 65536, (37, 53)).astype(np.uint16)
    a8 = rng.integers(0, 256, (20, 31)).astype(np.uint8)
    p = str(tmp_path / 't.tiff')
    for arr in (a16, a8):
        tiffio.write_tiff(p, arr)
        stack = np.full((3,) + arr.shape, 7, arr.dtype)
        assert tiffio.read_image_into(p, stack[1])
        np.testing.assert_array_equal(stack[1], arr)
        assert (stack[0] == 7).all() and (stack[2] == 7).all()
        for rows in (1, 8, arr.shape[0]):
            with open(p, 'wb') as fh:
                fh.write(_strip_tiff(arr, rows))
            out = np.zeros_like(arr)
            assert tiffio.read_image_into(p, out)
            np.testing.assert_array_equal(out, arr)
            np.testing.assert_array_equal(tiffio.read_image(p), arr)
    out = np.full_like(a16, 9)
    with open(p, 'wb') as fh:
        fh.write(_strip_tiff(a16, 8, big_endian=True))
    assert not tiffio.read_image_into(p, out) and (out == 9).all()          # byte order: needs a swap
    np.testing.assert_array_equal(tiffio.read_image(p), a16)
    with open(p, 'wb') as fh:
        fh.write(_strip_tiff(a16, 8, gap=6))
    assert not tiffio.read_image_into(p, out) and (out == 9).all()          # strips not back to back
    np.testing.assert_array_equal(tiffio.read_image(p), a16)
    tiffio.write_tiff(p, a16)
    assert not tiffio.read_image_into(p, np.zeros((37, 54), np.uint16))     # another shape
    assert not tiffio.read_image_into(p, np.zeros((37, 53), np.uint8))      # another dtype
    assert not tiffio.read_image_into(p, np.zeros((37, 106), np.uint16)[:, ::2])   # not contiguous
    tiffio.write_tiff(p, rng.integers(0, 256, (12, 9, 3)).astype(np.uint8))
    rgb = np.zeros((12, 9, 3), np.uint8)
    assert tiffio.read_image_into(p, rgb)
    np.testing.assert_array_equal(rgb, tiffio.read_image(p))
    with open(p, 'wb') as fh:
        fh.write(_strip_tiff(a16, 8)[:-900])                                  # pixel data cut short? no: the IFD is
    assert not tiffio.read_image_into(p, np.zeros_like(a16))                # gone -> not decodable here
    assert not tiffio.read_image_into(str(tmp_path / 'x.png'), np.zeros_like(a16))


def test_omezarr_roundtrip(tmp_path):
    """Store layout and chunk writing; the pyramid levels are inputs here (the product computes them on
    the device, the oracle stands in for it on CPU)."""
    from oracle import stitch_oracle as O
    rng = np.random.default_rng(1)
    img = rng.integers(0, 65536, (1, 2, 3, 701, 530)).astype(np.uint16)
    img[0, 1, 2] = 0    # an all-zero plane is not written (fill_value)
    levels = O.pyramid_nearest(img, 3)
    for compression in ('zlib', 'none'):
        path = str(tmp_path / f'r_{compression}.ome.zarr')
        shapes = omezarr.create_store(path, img.shape, img.dtype, pixel_size_um=0.5, dz_um=1.5,
                                      channel_names=['a 405', 'b 488'], channel_colors=[0xFF, 0xFF00], num_levels=3,
                                      compression=compression)
        assert shapes == [lv.shape for lv in levels] == [(1, 2, 3, 701, 530), (1, 2, 3, 350, 265), (1, 2, 3, 175, 132)]
        coords = [(0, c, z) for c in range(2) for z in range(3)]
        # two "processes" write different planes of the same store
        for part in (slice(0, 4), slice(4, 6)):
            n = omezarr.write_plane_levels(path, [lv.reshape((-1,) + lv.shape[3:])[part] for lv in levels], coords[part],
                                           compression=compression)
            assert n > 0
        for lv in range(3):
            np.testing.assert_array_equal(omezarr.read_array(os.path.join(path, str(lv))), levels[lv])
        assert not os.path.exists(os.path.join(path, '0', '0', '1', '2'))
        with open(os.path.join(path, '.zattrs')) as fh:
            attrs = json.load(fh)
        ms = attrs['multiscales'][0]
        assert [a['name'] for a in ms['axes']] == ['t', 'c', 'z', 'y', 'x']
        assert ms['datasets'][2]['coordinateTransformations'][0]['scale'] == [1, 1, 1.5, 2.0, 2.0]
        assert [c['label'] for c in attrs['omero']['channels']] == ['a 405', 'b 488']
    # single-level numpy input needs no GPU; a pyramid of a numpy input does (no host decimation exists)
    path = omezarr.write_ome_zarr(str(tmp_path / 'one.ome.zarr'), img, pixel_size_um=0.5, num_levels=1)
    np.testing.assert_array_equal(omezarr.read_array(os.path.join(path, '0')), img)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            omezarr.write_ome_zarr(str(tmp_path / 'two.ome.zarr'), img, pixel_size_um=0.5, num_levels=2)
    with pytest.raises(ValueError):
        omezarr.create_store(str(tmp_path / 'bad.ome.zarr'), img.shape, img.dtype, pixel_size_um=1.0, compression='lz4')


def test_ometiff_roundtrip(tmp_path):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 65536, (2, 3, 2, 37, 53)).astype(np.uint16)
    path = ometiff.write_ome_tiff(str(tmp_path / 'r.ome.tiff'), img, pixel_size_um=0.5, dz_um=1.5,
                                  channel_names=['a 405', 'b <488>', 'c'], channel_colors=[0xFF, 0xFF00, 0xFF0000], name='R0_t0')
    planes, xml = ometiff.read_ome_tiff(path)
    assert len(planes) == 12
    np.testing.assert_array_equal(np.stack(planes).reshape(img.shape), img)       # IFD order = T, C, Z (XYZCT)
    import xml.etree.ElementTree as ET
    root = ET.fromstring(xml)
    ns = {'o': 'http://www.openmicroscopy.org/Schemas/OME/2016-06'}
    px = root.find('o:Image/o:Pixels', ns)
    assert (px.get('SizeT'), px.get('SizeC'), px.get('SizeZ'), px.get('SizeY'), px.get('SizeX')) == ('2', '3', '2', '37', '53')
    assert px.get('DimensionOrder') == 'XYZCT' and px.get('Type') == 'uint16' and px.get('PhysicalSizeZ') == '1.5'
    assert [c.get('Name') for c in px.findall('o:Channel', ns)] == ['a 405', 'b <488>', 'c']
    assert len(px.findall('o:TiffData', ns)) == 12
    # the plain TIFF reader of this repo opens the first plane of the BigTIFF? no -- BigTIFF is not baseline
    with pytest.raises(ValueError):
        ometiff.write_ome_tiff(str(tmp_path / 'bad.ome.tiff'), img[0], pixel_size_um=1.0)


@pytest.mark.parametrize('name', REGION_CASES)
def test_placement_matches_golden_and_oracle(name):
    """Product geometry (placement.py) == reference placements recorded in the fixtures."""
    info, arrays = load_case(name)
    spec = spec_of(info)
    p = info['params']
    shifts = placement.Shifts()
    if p['use_registration']:
        shifts = placement.Shifts(tuple(info['h_shift']), tuple(info['v_shift']),
                                  tuple(info['h_shift_rev']) if 'h_shift_rev' in info else None,
                                  info.get('h_shift_rev_odd', 0))
    for key, cinfo in info['canvases'].items():
        region = key[1:].split('_', 1)[1]
        rows, cols = spec.dims_of(region)                                         # wells may hold different grids
        cells = [(r, c) for r in range(rows) for c in range(cols)]                # every FOV: positions may be off the grid
        xs = sorted(set(spec.stage_mm(r, c)[0] for r, c in cells))
        ys = sorted(set(spec.stage_mm(r, c)[1] for r, c in cells))
        wc, hc = placement.canvas_size(cols, rows, spec.tile_w, spec.tile_h,
                                       use_registration=p['use_registration'], shifts=shifts, xs=xs, ys=ys,
                                       pixel_size_um=spec.pixel_size_um)
        assert cinfo['shape'][3:] == [hc, wc]
        gold = arrays[f'{key}_placements']          # (c, z, x_pixel, y_pixel, row, col) BEFORE the crop offset
        for c, z, x_px, y_px, row, col in gold:
            if p['use_registration']:
                sy, sx, h, w, dy, dx = placement.registered_rect(int(row), int(col), rows, cols,
                                                                 spec.tile_w, spec.tile_h, shifts)
                assert (dy - sy, dx - sx) == (y_px, x_px)
        if not p['use_registration']:
            # coordinate mode: rebuild from stage positions in sorted-filename order
            fovs = [spec.fov_index(r, c, cols) for r, c in cells]
            order = placement.filename_order(fovs)
            per_plane = gold[(gold[:, 0] == 0) & (gold[:, 1] == 0)]
            for k, i in enumerate(order):
                r, c = divmod(i, cols)
                x_mm, y_mm = spec.stage_mm(r, c)
                rect = placement.coordinate_rect(x_mm, y_mm, min(xs), min(ys), spec.tile_w, spec.tile_h, spec.pixel_size_um)
                assert (rect[5], rect[4]) == (per_plane[k][2], per_plane[k][3])


def test_crop_origins_and_widths():
    spec = synth.GridSpec(rows=2, cols=2, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244)
    xs = [spec.stage_mm(0, c)[0] for c in range(2)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(2)]
    assert placement.registration_crop_widths(xs, ys, 2048, 2048, spec.pixel_size_um, 2) == \
        O.max_overlaps(xs, ys, 2048, 2048, spec.pixel_size_um, 2) == (256, 256)
    assert placement.horizontal_crop_origins(2048, 2048, 256) == (1024, 256, (512, 1792), (512, 0))
    assert placement.vertical_crop_origins(2048, 2048, 256) == (256, 1024, (1792, 512), (0, 512))
    a = np.arange(100 * 90).reshape(100, 90)
    n0, n1, (ry, rx), (my, mx) = placement.horizontal_crop_origins(100, 90, 30)
    ra, rb = O.overlap_crops_horizontal(a, a, 30)
    np.testing.assert_array_equal(a[ry:ry + n0, rx:rx + n1], ra)
    np.testing.assert_array_equal(a[my:my + n0, mx:mx + n1], rb)
    n0, n1, (ry, rx), (my, mx) = placement.vertical_crop_origins(100, 90, 30)
    ra, rb = O.overlap_crops_vertical(a, a, 30)
    np.testing.assert_array_equal(a[ry:ry + n0, rx:rx + n1], ra)
    np.testing.assert_array_equal(a[my:my + n0, mx:mx + n1], rb)
    with pytest.raises(ValueError, match='same shape'):
        placement.horizontal_crop_origins(100, 90, 0)      # [-0:] vs [:0]: the reference fails inside skimage


def test_stitcher_refuses_to_run_without_gpu(tmp_path):
    import torch
    from image_stitcher_amd.stitcher import Stitcher
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    spec = synth.GridSpec(rows=2, cols=2, tile_h=32, tile_w=32, ov_y=8, ov_x=8)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    st = Stitcher(StitchingParameters(input_folder=root))
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    assert st.regions == ['R0'] and st.num_z == 1 and (st.input_height, st.input_width) == (32, 32)
    xs = [spec.stage_mm(0, c)[0] for c in range(2)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(2)]
    assert st.calculate_output_dimensions(0, 'R0') == O.output_dimensions(xs, ys, 32, 32, spec.pixel_size_um, False)[:2]
    with pytest.raises(RuntimeError, match='no CPU path'):
        st.stitch_region(0, 'R0')


def test_grid_rects_vectorised_equals_per_tile_formula():
    rng = np.random.default_rng(0)
    for _ in range(60):
        n_rows, n_cols = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        w, h = int(rng.integers(20, 300)), int(rng.integers(20, 300))
        sh = placement.Shifts((int(rng.integers(-9, 10)), int(rng.integers(-w + 1, 5))),
                              (int(rng.integers(-h + 1, 5)), int(rng.integers(-9, 10))),
                              (int(rng.integers(-9, 10)), int(rng.integers(-w + 1, 5))) if rng.random() < 0.5 else None,
                              int(rng.integers(0, 2)))
        order = [(int(r), int(c)) for r, c in zip(rng.integers(0, n_rows, 9), rng.integers(0, n_cols, 9))]
        for crop in (True, False):
            for od in (None, order):
                want = np.array([placement.registered_rect(r, c, n_rows, n_cols, w, h, sh, crop)
                                 for r, c in (od or [(r, c) for r in range(n_rows) for c in range(n_cols)])])
                np.testing.assert_array_equal(placement.grid_rects(n_rows, n_cols, w, h, sh, od, crop), want.reshape(-1, 6))


def test_get_flatfields_estimators_and_basicpy_delegation(tmp_path, monkeypatch):
    """Flatfield *estimation* (stitcher.py:365-419): with basicpy present the reference's own call is made; without
    it the device restatement of BaSiC runs -- which needs the GPU, so on a CPU-only box it fails LOUDLY instead of
    silently dividing by something else; the smoothed mean exists only on explicit request.  uint8 acquisition with
    an RGB channel: one gain image per monochrome channel."""
    import sys
    import types
    from image_stitcher_amd.stitcher import Stitcher
    spec = synth.GridSpec(rows=2, cols=3, tile_h=32, tile_w=48, ov_y=8, ov_x=8, seed=3, dtype='uint8',
                          channels=('BF LED matrix full_RGB', 'Fluorescence 488 nm Ex'), rgb_channels=('BF LED matrix full_RGB',))
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)

    def prepared(**kw):
        st = Stitcher(StitchingParameters(input_folder=root, apply_flatfield=True), **kw)
        st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
        return st

    import torch
    monkeypatch.setitem(sys.modules, 'basicpy', None)            # import fails
    with pytest.raises(ImportError):
        prepared(flatfield_estimator='basicpy').get_flatfields()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no CPU path'):   # 'auto' -> the device BaSiC: no silent stand-in
            prepared().get_flatfields()
    with pytest.raises(ValueError):
        prepared(flatfield_estimator='bogus')
    st = prepared(flatfield_estimator='mean')                    # explicit request only
    seen = []
    st.get_flatfields(progress_callback=lambda i, n: seen.append((i, n)))
    assert st.flatfield_estimator_used == 'mean' and st.flatfield_info['estimator'] == 'mean'
    assert sorted(st.flatfields) == list(range(st.num_c)) == [0, 1, 2, 3] and len(seen) == 4
    for ff in st.flatfields.values():
        assert ff.shape == (32, 48) and ff.dtype == np.float32 and abs(float(ff.mean()) - 1.0) < 1e-5

    calls = []

    class FakeBaSiC:
        def __init__(self, **kw):
            calls.append(kw)

        def fit(self, images):
            assert images.ndim == 3 and images.shape[1:] == (32, 48)
            self.flatfield = np.full(images.shape[1:], 1.25, dtype=np.float64)

    monkeypatch.setitem(sys.modules, 'basicpy', types.SimpleNamespace(BaSiC=FakeBaSiC))
    st = prepared()
    st.flatfields[3] = np.ones((32, 48), np.float32)             # supplied by the caller: left alone
    st.get_flatfields()
    assert st.flatfield_estimator_used == 'basicpy'
    assert calls == [dict(get_darkfield=False, smoothness_flatfield=1)] * 3
    assert all(float(st.flatfields[i][0, 0]) == 1.25 for i in range(3)) and float(st.flatfields[3][0, 0]) == 1.0


def test_flatfield_sample_can_reach_80_images_and_the_device_fit_takes_them(tmp_path, monkeypatch):
    """the reference adds min(32, n) shuffled tiles per timepoint and stops once it holds MORE than 48
    (stitcher.py:381-395).  24 tiles per timepoint -> 24, 48, 72 images (48 is not > 48); the most it can reach is
    48 + 32 = 80.  The device fit's workspace accepts up to 80 and refuses 81."""
    import sys
    import types
    from image_stitcher_amd import native
    from image_stitcher_amd.stitcher import Stitcher
    spec = synth.GridSpec(rows=4, cols=6, tile_h=32, tile_w=48, ov_y=8, ov_x=8, seed=5, nt=4,
                          channels=('Fluorescence 488 nm Ex',))
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    counts = []

    class FakeBaSiC:
        def __init__(self, **kw):
            pass

        def fit(self, images):
            counts.append(len(images))
            self.flatfield = np.ones(images.shape[1:], dtype=np.float64)

    monkeypatch.setitem(sys.modules, 'basicpy', types.SimpleNamespace(BaSiC=FakeBaSiC))
    st = Stitcher(StitchingParameters(input_folder=root, apply_flatfield=True))
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    st.get_flatfields()
    assert counts == [72]                     # 24 + 24 + 24: the fourth timepoint is not touched
    L = native.lib()
    assert L.sq_basic_workspace_bytes(72, 32, 48) > 0 and L.sq_basic_workspace_bytes(80, 2048, 2048) > 0
    assert L.sq_basic_workspace_bytes(81, 32, 48) < 0 and b'1..80' in L.sq_last_error()


@pytest.mark.parametrize('rows,cols', [(1, 3), (3, 1), (1, 1)])
def test_degenerate_grids_behave_like_the_reference(tmp_path, rows, cols):
    """One row, one column or one tile: coordinate-only placement works; with -r the reference indexes
    x_pos_list[1] / y_pos_list[1] (stitcher.py:444-445) and dies with IndexError -- so does the drop-in."""
    from image_stitcher_amd.stitcher import Stitcher
    from image_stitcher_amd.tiffio import read_image
    from oracle import stitch_oracle as O
    spec = synth.GridSpec(rows=rows, cols=cols, tile_h=64, tile_w=96, ov_y=16, ov_x=24, seed=21)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    acq = O.parse_acquisition(root, read_image)
    want = O.stitch_region(acq, 0, 'R0', read_image, False, None)
    st = Stitcher(StitchingParameters(input_folder=root))
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    assert st.calculate_output_dimensions(0, 'R0') == (want.shape[-1], want.shape[-2])
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True))
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    with pytest.raises(IndexError):
        st.calculate_shifts(0, 'R0')
    with pytest.raises(IndexError):
        O.calculate_shifts(acq, 0, 'R0', read_image, '', 0, 'Unidirectional', 'phase')


def test_row_bands_and_plane_band_units():
    from image_stitcher_amd import omezarr, sharding
    assert sharding.row_bands(4343, 3) == [(0, 2048), (2048, 4096), (4096, 4343)]
    assert sharding.row_bands(4343, 1) == [(i, min(i + 512, 4343)) for i in range(0, 4343, 512)]
    assert sharding.row_bands(300, 2) == [(0, 300)] and sharding.row_bands(1, 5) == [(0, 1)]
    for h, levels in ((4343, 3), (36473, 6), (73193, 7), (512, 1), (513, 1)):
        bands = sharding.row_bands(h, levels)
        assert bands[0][0] == 0 and bands[-1][1] == h and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
        for y0, _ in bands:
            for lv in range(levels):      # a band starts on a chunk row of every level
                assert (y0 >> lv) % min(512, max(1, h >> lv)) == 0
    # enough planes: whole planes; fewer planes than ranks: (plane, band) units, each exactly once
    assert sharding.plane_band_units(4, [(0, 10), (10, 20)], 1, 2) == [(2, -1), (3, -1)]
    seen = sorted(u for r in range(3) for u in sharding.plane_band_units(1, [(0, 1), (1, 2), (2, 3), (3, 4)], r, 3))
    assert seen == [(0, b) for b in range(4)]
    assert sharding.plane_band_units(1, [(0, 9)], 0, 4) == [(0, -1)] and sharding.plane_band_units(1, [(0, 9)], 1, 4) == []
    # rectangles cut to a band
    assert placement.clip_rect_to_rows((10, 5, 100, 50, 200, 7), 0, 250) == (10, 5, 50, 50, 200, 7)
    assert placement.clip_rect_to_rows((10, 5, 100, 50, 200, 7), 250, 400) == (60, 5, 50, 50, 0, 7)
    assert placement.clip_rect_to_rows((10, 5, 100, 50, 200, 7), 220, 260) == (30, 5, 40, 50, 0, 7)
    assert placement.clip_rect_to_rows((10, 5, 100, 50, 200, 7), 300, 400) is None
    assert placement.clip_rect_to_rows((10, 5, 100, 50, 200, 7), 0, 200) is None
    # the chunk jobs of a band land on the chunk rows of the full levels
    full = [np.arange(1100 * 600, dtype=np.uint16).reshape(1, 1100, 600), np.zeros((1, 550, 300), np.uint16)]
    band = [full[0][:, 1024:], full[1][:, 512:]]
    jobs = omezarr.chunk_jobs(band, [(0, 0, 0)], (1, 1, 1, 512, 512), row_offset=1024, level_heights=[1100, 550])
    assert sorted((j[0], j[4], j[5]) for j in jobs) == [(0, 2, 0), (0, 2, 1), (1, 1, 0)]
    with pytest.raises(ValueError):
        omezarr.chunk_jobs(band, [(0, 0, 0)], (1, 1, 1, 512, 512), row_offset=1000, level_heights=[1100, 550])


def test_dynamic_registration_is_stored_and_ignored_like_the_reference(tmp_path, monkeypatch):
    """The reference parses --dynamic-registration, stores it (stitcher.py:92) and never reads it: with the flag set its
    shifts are the centre-pair ones.  Same here -- on the golden acquisition whose centre tile is blank (where the
    all-pairs median of this build WOULD differ) ``dynamic_registration=True`` ends with the reference's golden shifts and
    never enters the all-pairs path; that path sits behind a flag of this build (--all-pairs-registration).  No GPU here:
    the two pair registrations go through the oracle (the device versions are pinned by tests/test_stitcher_gpu.py)."""
    from image_stitcher_amd import registration
    from image_stitcher_amd.stitcher import Stitcher
    info, _ = load_case('reg_blank_centre')
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec_of(info), root)
    monkeypatch.setattr(registration, 'check_crop_lengths', lambda *a, **k: None)     # asks the library; not under test
    a = stitcher_cli.parse_args(['-i', root, '-r', '--dynamic-registration'])
    assert a.dynamic_registration and not a.all_pairs_registration
    st = Stitcher(stitcher_cli.create_params(a), normalization=None, all_pairs_registration=a.all_pairs_registration)
    assert st.dynamic_registration is True and st.all_pairs_registration is False
    st.calculate_horizontal_shift = lambda l, r, w: O.calculate_horizontal_shift(l, r, w, st.dtype, None)
    st.calculate_vertical_shift = lambda t, b, w: O.calculate_vertical_shift(t, b, w, st.dtype, None)
    st._calculate_shifts_all_pairs = lambda *a, **k: pytest.fail("dynamic_registration must not select all-pairs registration")
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    st.calculate_shifts(st.timepoints[0], st.regions[0])
    assert (list(st.h_shift), list(st.v_shift)) == (info['h_shift'], info['v_shift'])
    # and the flag of this build does select it
    b = stitcher_cli.parse_args(['-i', root, '-r', '--all-pairs-registration'])
    st2 = Stitcher(stitcher_cli.create_params(b), normalization=None, all_pairs_registration=b.all_pairs_registration)
    assert st2.all_pairs_registration and not st2.dynamic_registration
    called = []
    st2._calculate_shifts_all_pairs = lambda *a, **k: called.append(1)
    st2.get_timepoints(); st2.extract_acquisition_parameters(); st2.get_pixel_size(); st2.parse_acquisition_metadata()
    st2.calculate_shifts(st2.timepoints[0], st2.regions[0])
    assert called == [1]
    # without -r the extension is off whatever the flag says
    assert not Stitcher(StitchingParameters(input_folder=root), all_pairs_registration=True).all_pairs_registration


def test_native_chunk_file_writer(tmp_path):
    """sq_write_files (the chunk files of the OME-Zarr store, written by native threads): every file holds exactly its byte range,
    empty ranges make empty files, existing files are replaced, a missing directory is an error that names the path."""
    from image_stitcher_amd import native
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, size=1 << 20, dtype=np.uint8)
    cuts = np.sort(rng.choice(np.arange(1, data.size), size=400, replace=False))
    offsets = np.concatenate([[0], cuts, cuts[-1:], [data.size]]).astype(np.int64)      # 402 files, one of them empty
    for d in range(7):
        os.makedirs(tmp_path / str(d))
    paths = [str(tmp_path / str(i % 7) / f'chunk.{i}') for i in range(len(offsets) - 1)]
    with open(paths[5], 'wb') as fh:
        fh.write(b'x' * 100000)                                                          # longer than what replaces it
    assert native.write_files(paths, data, offsets, n_threads=8) == data.size
    for i, p in enumerate(paths):
        with open(p, 'rb') as fh:
            assert fh.read() == data[offsets[i]:offsets[i + 1]].tobytes(), i
    assert os.path.getsize(paths[-2]) == 0
    assert native.write_files([], data, np.zeros(1, np.int64)) == 0
    with pytest.raises(native.NativeError, match='no_such_dir'):
        native.write_files([str(tmp_path / 'no_such_dir' / 'x')], data, np.array([0, 10]))
    with pytest.raises(ValueError):
        native.write_files(paths[:2], data, np.array([0, 10]))                           # offsets do not match the paths

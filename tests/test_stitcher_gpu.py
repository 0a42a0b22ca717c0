"""End-to-end parity: the drop-in ``Stitcher`` on the GPU against canvases, shifts and
placements produced by the unmodified reference (tests/golden/*.npz|json)."""
import os

import numpy as np
import pytest

from helpers import PHASE_MAY_DIFFER, REGION_CASES, flatfields_for, load_case, sha, spec_of
from image_stitcher_amd import omezarr, synth
from image_stitcher_amd.stitcher import Stitcher
from image_stitcher_amd.stitcher_parameters import StitchingParameters
from image_stitcher_amd import stitcher_cli

pytestmark = pytest.mark.gpu


def _prepared(info, root, normalization):
    p = info['params']
    params = StitchingParameters(input_folder=root, use_registration=p['use_registration'],
                                 apply_flatfield=p['apply_flatfield'],
                                 registration_channel=p['registration_channel'],
                                 registration_z_level=p['registration_z_level'],
                                 scan_pattern=info['spec']['scan_pattern'])
    st = Stitcher(params, normalization=normalization)
    st.get_timepoints()
    st.extract_acquisition_parameters()
    st.get_pixel_size()
    st.parse_acquisition_metadata()
    flats = flatfields_for(info, st.num_c)
    if flats:
        st.flatfields = flats
    return st


@pytest.mark.parametrize('name', REGION_CASES)
def test_stitcher_matches_reference(name, tmp_path):
    info, arrays = load_case(name)
    spec = spec_of(info)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    st = _prepared(info, root, normalization=None)     # golden shifts come from scikit-image 0.18.3
    assert st.regions == info['regions'] and st.monochrome_channels == info['channels']
    assert st.num_z == info['num_z'] and str(np.dtype(st.dtype)) == info['dtype']
    if info.get('forced'):      # dictated shifts (every sign combination): placement and fusion only
        st.h_shift, st.v_shift = tuple(info['h_shift']), tuple(info['v_shift'])
        if 'h_shift_rev' in info:
            st.h_shift_rev, st.h_shift_rev_odd = tuple(info['h_shift_rev']), info['h_shift_rev_odd']
    elif info['params']['use_registration']:
        st.calculate_shifts(st.timepoints[0], st.regions[0])
        assert list(st.h_shift) == info['h_shift'] and list(st.v_shift) == info['v_shift']
        if spec.scan_pattern == 'S-Pattern':
            assert list(st.h_shift_rev) == info['h_shift_rev']
            assert int(st.h_shift_rev_odd) == info['h_shift_rev_odd']
        # the scikit-image >= 0.19 default recovers the same integers on these scenes
        st2 = _prepared(info, root, normalization='phase')
        st2.calculate_shifts(st2.timepoints[0], st2.regions[0])
        if name in PHASE_MAY_DIFFER:     # ... but there the device agrees with the oracle's phase mode
            from image_stitcher_amd.tiffio import read_image
            from oracle import stitch_oracle as O
            acq = O.parse_acquisition(root, read_image)
            ph = O.calculate_shifts(acq, acq.timepoints[0], acq.regions[0], read_image, info['params']['registration_channel'],
                                    info['params']['registration_z_level'], spec.scan_pattern, normalization='phase')
            assert (tuple(st2.h_shift), tuple(st2.v_shift)) == (tuple(ph['h_shift']), tuple(ph['v_shift']))
        else:
            assert (list(st2.h_shift), list(st2.v_shift)) == (info['h_shift'], info['v_shift'])
    for key, cinfo in info['canvases'].items():
        t, region = key[1:].split('_', 1)
        canvas = st.stitch_region(int(t), region)
        assert list(canvas.shape) == cinfo['shape']
        assert st.num_pyramid_levels == cinfo['num_pyramid_levels']
        assert sha(canvas) == cinfo['sha256']
        if f'{key}_canvas' in arrays:
            np.testing.assert_array_equal(canvas, arrays[f'{key}_canvas'])
        for wi, (c, z, y0, x0, hh, ww) in enumerate(cinfo.get('windows', [])):
            np.testing.assert_array_equal(canvas[0, c, z, y0:y0 + hh, x0:x0 + ww], arrays[f'{key}_win{wi}'])


def test_single_tile_methods_match_golden(tmp_path):
    v = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'flatfield_vectors.npz'))
    spec = synth.GridSpec(rows=1, cols=1, tile_h=48, tile_w=64, ov_y=0, ov_x=0, seed=77)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    st = Stitcher(StitchingParameters(input_folder=root, apply_flatfield=True))
    for dt in ('float32', 'float64'):
        st.flatfields = {0: v[f'ff_{dt}']}
        np.testing.assert_array_equal(st.apply_flatfield_correction(v['tile'], 0), v[f'out_{dt}'])
    tile = v['tile']
    assert st.apply_flatfield_correction(tile, 3) is tile
    n = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'normalize_vectors.npz'))
    np.testing.assert_array_equal(st.normalize_image(n['in_uint16']), n['out_uint16'])


def test_cli_end_to_end_writes_ome_zarr(tmp_path, capsys):
    info, arrays = load_case('reg_3x4_small')
    spec = spec_of(info)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    stitcher_cli.main(['-i', root, '-r', '--registration-channel', info['params']['registration_channel'],
                       '--registration-z-level', '1', '--normalization', 'none'])
    out_dirs = [d for d in os.listdir(tmp_path) if d.startswith('acq_stitched_')]
    assert len(out_dirs) == 1
    store = os.path.join(tmp_path, out_dirs[0], '0_stitched', 'R0_stitched.ome.zarr')
    got = omezarr.read_array(os.path.join(store, '0'))
    np.testing.assert_array_equal(got, arrays['t0_R0_canvas'])
    # bad input folder: message on stderr, exit code 1 (stitcher_cli.py:114-116)
    with pytest.raises(SystemExit) as e:
        stitcher_cli.main(['-i', str(tmp_path / 'missing')])
    assert e.value.code == 1
    assert 'Error: Input folder does not exist' in capsys.readouterr().err


def test_feather_mode_runs_and_agrees_with_overwrite_away_from_seams(tmp_path):
    info, arrays = load_case('reg_neg_skew')
    spec = spec_of(info)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    st = _prepared(info, root, 'phase')
    st.fusion_mode = 'feather'
    st.calculate_shifts(0, 'R0')
    canvas = st.stitch_region(0, 'R0')
    ref = arrays['t0_R0_canvas']
    assert canvas.shape == ref.shape
    # tile interiors (covered by one tile only) are identical in both modes
    np.testing.assert_array_equal(canvas[0, 0, 0, 50:80, 50:80], ref[0, 0, 0, 50:80, 50:80])


def test_run_fires_the_reference_signals(tmp_path):
    """run() (what stitcher_cli calls, and what a QThread front-end would start) fires the reference's signals
    (stitcher.py:33-37) in its order."""
    info, arrays = load_case('reg_neg_skew')
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec_of(info), root)
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization=None)
    seen = {'progress': 0, 'started': 0, 'saving': [], 'finished': None}
    st.update_progress.connect(lambda a, b: seen.__setitem__('progress', seen['progress'] + 1))
    st.starting_stitching.connect(lambda: seen.__setitem__('started', seen['started'] + 1))
    st.starting_saving.connect(lambda final: seen['saving'].append(final))
    st.finished_saving.connect(lambda path, dtype: seen.__setitem__('finished', (path, dtype)))
    st.run()
    assert seen['started'] == 1 and seen['progress'] == 9 and seen['saving'] == [False, True]
    path, dtype = seen['finished']
    assert path.endswith(os.path.join('0_stitched', 'R0_stitched.ome.zarr')) and dtype == np.uint16
    np.testing.assert_array_equal(omezarr.read_array(os.path.join(path, '0')), arrays['t0_R0_canvas'])
    assert (list(st.h_shift), list(st.v_shift)) == (info['h_shift'], info['v_shift'])


def test_all_pairs_registration_survives_a_bad_centre_tile(tmp_path):
    """--all-pairs-registration (an addition of this build) = all adjacent pairs, median; --dynamic-registration is
    stored and ignored like in the reference (stitcher.py:92) and leaves the centre-pair result.  With the centre
    tile replaced by an empty field the centre-pair result is what the reference computes for a blank tile
    (golden case reg_blank_centre) -- wrong for the mosaic; the all-pairs median is still the planted drift.
    4 x 5 S-Pattern grid: the blank tile spoils 2 of the 8 pairs of its row group, a minority."""
    from image_stitcher_amd.tiffio import write_tiff
    spec = synth.GridSpec(rows=4, cols=5, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=16,
                          scan_pattern='S-Pattern', rev_ov_x=48, rev_jy=-2)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)

    def shifts(all_pairs, dynamic=False):
        st = Stitcher(StitchingParameters(input_folder=root, use_registration=True, dynamic_registration=dynamic,
                                          scan_pattern='S-Pattern'), normalization='phase', all_pairs_registration=all_pairs)
        st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
        st.calculate_shifts(st.timepoints[0], st.regions[0])
        return tuple(st.h_shift), tuple(st.v_shift), tuple(st.h_shift_rev), int(st.h_shift_rev_odd)

    centre = shifts(False)
    assert centre[0] == (-2, -48) and centre[2] == (3, -44)       # centre row 1 is a reversed row here
    dyn = shifts(True)
    assert dyn[0] == centre[0] and dyn[2] == centre[2] and dyn[3] == centre[3]
    # (v_shift: this synthetic S-Pattern has two populations of vertical pairs -- rows of different parity
    #  have different pitches by construction -- so the median legitimately differs from the centre pair)
    # wipe the centre tile (row 1, col 2 -> fov of the reversed row)
    centre_fov = spec.fov_index(1, 2)
    path = os.path.join(root, '0', f'R0_{centre_fov}_0_{synth.channel_file_token(spec.channels[0])}.tiff')
    write_tiff(path, np.full((spec.tile_h, spec.tile_w), 1234, dtype=np.uint16))
    derailed = shifts(False)
    assert derailed[0] != centre[0]               # the reference's centre-pair scheme is derailed
    assert shifts(False, dynamic=True) == derailed    # ... and its --dynamic-registration flag changes nothing, as in the reference
    after = shifts(True)                          # the all-pairs median is not (v_shift: two populations, see above)
    assert (after[0], after[2], after[3]) == (dyn[0], dyn[2], dyn[3]) and after[1][0] == dyn[1][0]


def test_run_writes_shift_table_single_process(tmp_path):
    """run() leaves shift_table.json beside the stores: one entry per (timepoint, region) -- the shared
    shifts of the reference's register-once scheme, or each unit's own with per_region_registration."""
    import json
    info, arrays = load_case('reg_multi')
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec_of(info), root)
    for per_region in (False, True):
        st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization=None,
                      per_region_registration=per_region)
        st.run()
        with open(os.path.join(st.output_folder, 'shift_table.json')) as fh:
            table = json.load(fh)
        assert table['per_region_registration'] is per_region and len(table['shifts']) == 4
        assert all((e['h_shift'], e['v_shift']) == (info['h_shift'], info['v_shift']) for e in table['shifts'])
        for key in info['canvases']:
            t, region = key[1:].split('_', 1)
            store = os.path.join(st.output_folder, f'{t}_stitched', f'{region}_stitched.ome.zarr')
            np.testing.assert_array_equal(omezarr.read_array(os.path.join(store, '0')), arrays[f'{key}_canvas'])


@pytest.mark.parametrize('ext', ['bmp', 'png'])
def test_bmp_and_png_tiles(tmp_path, ext):
    """Squid also saves .bmp tiles (and the reference accepts .png): uint8 acquisition with registration,
    decoded through PIL, against the oracle on the same files."""
    from image_stitcher_amd.tiffio import read_image
    from oracle import stitch_oracle as O
    spec = synth.GridSpec(rows=2, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=17, dtype='uint8', noise=0, nz=2)
    root = str(tmp_path / 'acq')
    paths = synth.write_acquisition(spec, root, ext=ext)
    assert all(p.endswith('.' + ext) for p in paths)
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), normalization='phase')
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    assert st.dtype == np.uint8 and st.num_z == 2
    st.calculate_shifts(0, 'R0')
    acq = O.parse_acquisition(root, read_image)
    want_shifts = O.calculate_shifts(acq, 0, 'R0', read_image, '', 0, 'Unidirectional', 'phase')
    assert (st.h_shift, st.v_shift) == (want_shifts['h_shift'], want_shifts['v_shift'])
    np.testing.assert_array_equal(st.stitch_region(0, 'R0'), O.stitch_region(acq, 0, 'R0', read_image, True, want_shifts))


@pytest.mark.parametrize('rows,cols', [(1, 3), (3, 1), (1, 1)])
def test_degenerate_grids_coordinate_mode(tmp_path, rows, cols):
    """One row, one column, one tile (coordinate-only): fused canvas equals the oracle's."""
    from image_stitcher_amd.tiffio import read_image
    from oracle import stitch_oracle as O
    spec = synth.GridSpec(rows=rows, cols=cols, tile_h=64, tile_w=96, ov_y=16, ov_x=24, seed=21, nz=2)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    st = Stitcher(StitchingParameters(input_folder=root))
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    acq = O.parse_acquisition(root, read_image)
    np.testing.assert_array_equal(st.stitch_region(0, 'R0'), O.stitch_region(acq, 0, 'R0', read_image, False, None))


def test_shifts_the_reference_cannot_place_fail_the_same_way(tmp_path):
    """h_shift (-9, 15), v_shift (-14, -11): a tile then starts beyond the canvas edge, the reference's
    ``tile[:y_end - y, :x_end - x]`` wraps around and numpy refuses the assignment (make_golden.forced_cases
    records the ValueError).  Same input, same failure -- in the oracle and in the drop-in."""
    from image_stitcher_amd.tiffio import read_image
    from oracle import stitch_oracle as O
    spec = synth.GridSpec(rows=3, cols=4, tile_h=64, tile_w=80, ov_y=16, ov_x=20, seed=44)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    st = Stitcher(StitchingParameters(input_folder=root, use_registration=True))
    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
    st.h_shift, st.v_shift = (-9, 15), (-14, -11)
    with pytest.raises(ValueError, match=r"could not broadcast input array from shape \(61,70\) into shape \(61,0\)"):
        st.stitch_region(0, 'R0')
    acq = O.parse_acquisition(root, read_image)
    with pytest.raises(ValueError, match=r"could not broadcast input array from shape \(61,70\) into shape \(61,0\)"):
        O.stitch_region(acq, 0, 'R0', read_image, True, dict(h_shift=(-9, 15), v_shift=(-14, -11)))


def test_cli_with_ff_estimates_gains_on_the_device_and_says_so(tmp_path, capsys):
    """`-r -ff` through the CLI without basicpy: the gains come from the device restatement of the BaSiC fit (parity with
    basicpy unpinned, and the run says so on stdout, in flatfield_info.json and in shift_table.json); the canvas is the
    oracle's for THOSE gains (the divide is exact whatever the gains are)."""
    import json
    from image_stitcher_amd.tiffio import read_image
    from oracle import stitch_oracle as O
    spec = synth.GridSpec(rows=3, cols=4, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=77, channels=synth.DEFAULT_CHANNELS[:2])
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    params = StitchingParameters(input_folder=root, use_registration=True, apply_flatfield=True)
    st = Stitcher(params, normalization='phase', flatfield_estimator='basic')
    st.run()
    out = capsys.readouterr().out
    assert 'parity with it is unpinned' in out and st.flatfield_estimator_used == 'basic'
    with open(os.path.join(st.output_folder, 'flatfield_info.json')) as fh:
        info = json.load(fh)
    assert info['estimator'] == 'basic' and sorted(info['channels']) == sorted(st.channel_names)
    assert all(c['images'] == 12 and c['working_size'] == 128 and c['ladmap_iterations'] >= 1 for c in info['channels'].values())
    with open(os.path.join(st.output_folder, 'shift_table.json')) as fh:
        assert json.load(fh)['flatfield_estimator'] == 'basic'
    assert sorted(st.flatfields) == [0, 1]
    for ff in st.flatfields.values():
        assert ff.shape == (128, 160) and ff.dtype == np.float32 and abs(float(ff.mean()) - 1.0) < 0.05 and np.isfinite(ff).all()
    acq = O.parse_acquisition(root, read_image)
    shifts = {'h_shift': tuple(st.h_shift), 'v_shift': tuple(st.v_shift)}
    want = O.stitch_region(acq, 0, 'R0', read_image, True, shifts, st.flatfields, True)
    store = os.path.join(st.output_folder, '0_stitched', 'R0_stitched.ome.zarr')
    np.testing.assert_array_equal(omezarr.read_array(os.path.join(store, '0')), want)


@pytest.mark.parametrize('layout', ['strips_ifd_last', 'big_endian'])
def test_tile_files_in_other_tiff_layouts(tmp_path, layout):
    """The tile files of a golden acquisition rewritten the way libtiff lays them out (8-row strips, IFD after the pixel
    data: read straight into the staging buffer by ``tiffio.read_image_into``) and big-endian (declined there, decoded by
    ``read_image``): the same canvas as the reference's, voxel for voxel."""
    from test_host_cpu import _strip_tiff
    from image_stitcher_amd import tiffio
    info, arrays = load_case('reg_3x4_small')
    spec = spec_of(info)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    n = 0
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(('.tiff', '.tif')):
                p = os.path.join(dp, f)
                img = tiffio.read_image(p)
                with open(p, 'wb') as fh:
                    fh.write(_strip_tiff(img, 8, big_endian=(layout == 'big_endian')))
                probe = np.zeros_like(img)
                assert tiffio.read_image_into(p, probe) == (layout != 'big_endian')
                n += 1
    assert n >= 12
    st = _prepared(info, root, normalization=None)
    st.calculate_shifts(st.timepoints[0], st.regions[0])
    assert list(st.h_shift) == info['h_shift'] and list(st.v_shift) == info['v_shift']
    key, cinfo = next(iter(info['canvases'].items()))
    t, region = key[1:].split('_', 1)
    canvas = st.stitch_region(int(t), region)
    assert sha(canvas) == cinfo['sha256']

"""The oracle (oracle/stitch_oracle.py) against vectors produced by the real reference.

CPU only.  Everything here is bit-exact: integers, uint16/uint8 voxels, and the float
shifts (which are multiples of 0.1 by construction of upsample_factor=10)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, PHASE_MAY_DIFFER, REGION_CASES, flatfields_for, load_case, sha, spec_of
from image_stitcher_amd import synth
from image_stitcher_amd.tiffio import read_image
from oracle import stitch_oracle as O


@pytest.fixture(scope='module')
def pcc_cases():
    with open(os.path.join(GOLDEN, 'pcc_vectors.json')) as fh:
        return json.load(fh), np.load(os.path.join(GOLDEN, 'pcc_vectors.npz'))


def _pcc_inputs(case, i, arrays):
    """Rebuild the crops exactly as make_golden.pcc_vectors did (big ones are not stored)."""
    n0, n1 = case['shape']
    if f'ref{i}' in arrays:
        return arrays[f'ref{i}'], arrays[f'mov{i}']
    seed = case['seed']
    dy, dx = case['planted']
    big = synth.scene_patch(seed, 100, 100, n0 + 32, n1 + 32)
    ref = big[16:16 + n0, 16:16 + n1]
    mov = big[16 - dy:16 - dy + n0, 16 - dx:16 - dx + n1] + synth.noise_patch(seed + 1, n0, n1, 150)
    ref, mov = (O.normalize_image(a.astype(np.uint16), np.uint16) for a in (ref, mov))
    assert sha(ref) == case['ref_sha'] and sha(mov) == case['mov_sha']
    return ref, mov


def test_pcc_matches_skimage_018(pcc_cases):
    cases, arrays = pcc_cases
    for i, case in enumerate(cases):
        ref, mov = _pcc_inputs(case, i, arrays)
        s, err, ph, _ = O.phase_cross_correlation(ref, mov, 10, normalization=None)
        assert s.tolist() == case['shift_none'], case['shape']
        assert err == pytest.approx(case['error_none'], rel=1e-9, abs=1e-12)
        assert ph == pytest.approx(case['phasediff_none'], abs=1e-9)
        s1, _, _, _ = O.phase_cross_correlation(ref, mov, 1, normalization=None)
        assert s1.tolist() == case['shift_int']


def test_pcc_phase_mode_matches_driven_skimage(pcc_cases):
    cases, arrays = pcc_cases
    for i, case in enumerate(cases):
        ref, mov = _pcc_inputs(case, i, arrays)
        s, _, _, _ = O.phase_cross_correlation(ref, mov, 10, normalization='phase')
        assert s.tolist() == case['shift_phase'], case['shape']


def test_pcc_rejects_shape_mismatch_and_bad_mode():
    a = np.zeros((8, 8))
    with pytest.raises(ValueError, match="same shape"):
        O.phase_cross_correlation(a, np.zeros((8, 9)))
    with pytest.raises(ValueError, match="normalization"):
        O.phase_cross_correlation(a, a, normalization='bogus')


def test_normalize_image_golden():
    v = np.load(os.path.join(GOLDEN, 'normalize_vectors.npz'))
    for dt in ('uint16', 'uint8'):
        out = O.normalize_image(v[f'in_{dt}'], np.dtype(dt).type)
        assert out.dtype == np.dtype(dt)
        np.testing.assert_array_equal(out, v[f'out_{dt}'])
        # constant tile: 0/0 -> NaN -> 0 through the reference's cast (recorded from the reference itself)
        np.testing.assert_array_equal(O.normalize_image(v[f'const_in_{dt}'], np.dtype(dt).type), v[f'const_out_{dt}'])
        assert not v[f'const_out_{dt}'].any()


def test_flatfield_golden():
    v = np.load(os.path.join(GOLDEN, 'flatfield_vectors.npz'))
    for dt in ('float32', 'float64'):
        out = O.apply_flatfield(v['tile'], v[f'ff_{dt}'], np.uint16)
        np.testing.assert_array_equal(out, v[f'out_{dt}'])
    tile = v['tile']
    assert O.apply_flatfield(tile, None, np.uint16) is tile


@pytest.mark.parametrize('name', REGION_CASES)
def test_region_case(name, tmp_path):
    info, arrays = load_case(name)
    spec = spec_of(info)
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec, root)
    acq = O.parse_acquisition(root, read_image)
    p = info['params']
    assert acq.regions == info['regions']
    assert acq.monochrome_channels == info['channels']
    assert acq.num_z == info['num_z']
    assert str(np.dtype(acq.dtype)) == info['dtype']
    shifts = None
    if info.get('forced'):      # dictated shifts: the reference's integer geometry on arbitrary sign combinations
        shifts = dict(h_shift=tuple(info['h_shift']), v_shift=tuple(info['v_shift']))
        if 'h_shift_rev' in info:
            shifts.update(h_shift_rev=tuple(info['h_shift_rev']), h_shift_rev_odd=info['h_shift_rev_odd'])
    elif p['use_registration']:
        shifts = O.calculate_shifts(acq, acq.timepoints[0], acq.regions[0], read_image,
                                    p['registration_channel'], p['registration_z_level'],
                                    spec.scan_pattern, normalization=None)
        assert list(shifts['h_shift']) == info['h_shift']
        assert list(shifts['v_shift']) == info['v_shift']
        if spec.scan_pattern == 'S-Pattern':
            assert list(shifts['h_shift_rev']) == info['h_shift_rev']
            assert int(shifts['h_shift_rev_odd']) == info['h_shift_rev_odd']
        # phase mode recovers the same integers on these scenes
        ph = O.calculate_shifts(acq, acq.timepoints[0], acq.regions[0], read_image,
                                p['registration_channel'], p['registration_z_level'],
                                spec.scan_pattern, normalization='phase')
        if name not in PHASE_MAY_DIFFER:
            assert list(ph['h_shift']) == info['h_shift'] and list(ph['v_shift']) == info['v_shift']
    flats = flatfields_for(info, len(acq.monochrome_channels))
    grid_dim = 1
    if len(acq.regions) > 1:
        grid_dim = max(len(set(r[0] for r in acq.regions)), len(set(r[1:] for r in acq.regions)))
    for key, cinfo in info['canvases'].items():
        t, region = key[1:].split('_', 1)
        plan = O.plan_region(acq, t, region, p['use_registration'],
                             *( [shifts['h_shift'], shifts['v_shift'], shifts.get('h_shift_rev'),
                                 shifts.get('h_shift_rev_odd', 0)] if shifts else [(0, 0), (0, 0), None, 0]),
                             grid_dim)
        assert [1, len(acq.monochrome_channels), acq.num_z, plan.canvas_h, plan.canvas_w] == cinfo['shape']
        assert plan.levels == cinfo['num_pyramid_levels']
        gold_pl = arrays[f'{key}_placements']
        mine = np.array([[f['c'], f['z'], f['x_px'], f['y_px']] for f in plan.files])
        np.testing.assert_array_equal(mine, gold_pl[:, :4])
        canvas = O.stitch_region(acq, t, region, read_image, p['use_registration'], shifts, flats,
                                 p['apply_flatfield'])
        assert sha(canvas) == cinfo['sha256']
        if f'{key}_canvas' in arrays:
            np.testing.assert_array_equal(canvas, arrays[f'{key}_canvas'])
        for wi, (c, z, y0, x0, hh, ww) in enumerate(cinfo.get('windows', [])):
            np.testing.assert_array_equal(canvas[0, c, z, y0:y0 + hh, x0:x0 + ww], arrays[f'{key}_win{wi}'])


def test_pyramid_nearest_golden():
    """The pyramid restatement against third-party executions of the resize Scaler.nearest performs
    (make_golden.pyramid_vectors): scikit-image 0.18.3 for odd sizes, scipy's grid-mode zoom (the
    scikit-image >= 0.19 path) for all sizes."""
    v = np.load(os.path.join(GOLDEN, 'pyramid_vectors.npz'))
    n_sk = n_zoom = 0
    for key in v.files:
        if not key.startswith('in_'):
            continue
        i = key[3:]
        got = O.pyramid_nearest(v[key], 2)[1]
        np.testing.assert_array_equal(got, v[f'zoom_{i}'])
        n_zoom += 1
        if f'sk_{i}' in v.files:
            np.testing.assert_array_equal(got, v[f'sk_{i}'])
            n_sk += 1
    assert n_zoom >= 8 and n_sk >= 3


def test_pyramid_nearest_matches_live_scipy_zoom():
    """Same check against the scipy of this interpreter, on more sizes, over several levels."""
    ndi = pytest.importorskip('scipy.ndimage')
    rng = np.random.default_rng(5)
    for shape in [(2, 301, 212), (1, 64, 64), (3, 17, 255), (1, 1, 9), (1, 9, 1)]:
        img = rng.integers(0, 65536, shape).astype(np.uint16)
        levels = O.pyramid_nearest(img, 5)
        for lv in range(1, len(levels)):
            prev = levels[lv - 1]
            oy, ox = prev.shape[-2] // 2, prev.shape[-1] // 2
            assert levels[lv].shape == prev.shape[:-2] + (oy, ox)
            for pl in range(shape[0]):
                want = ndi.zoom(prev[pl], (oy / prev.shape[-2], ox / prev.shape[-1]), order=0, mode='mirror', grid_mode=True)
                np.testing.assert_array_equal(levels[lv][pl], want)
    assert len(O.pyramid_nearest(np.zeros((1, 1, 9), np.uint8), 5)) == 1       # a size-1 axis ends the pyramid

"""Parity of the HIP registration pipeline (through the C-ABI) against the golden vectors of
the real reference (skimage 0.18.3 = normalization None; phase mode via the driven
construction) and against the oracle on fresh seeded inputs.

Shifts are multiples of 1/upsample_factor by construction, so equality is exact; the bar in
BASELINE.json is +-0.5 px for sub-pixel shifts and bit-exact for the rounded integers."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, sha
from image_stitcher_amd import native, placement, registration, synth
from oracle import stitch_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _pcc_inputs(case, i, arrays):
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


def test_golden_pcc_vectors_both_modes():
    with open(os.path.join(GOLDEN, 'pcc_vectors.json')) as fh:
        cases = json.load(fh)
    arrays = np.load(os.path.join(GOLDEN, 'pcc_vectors.npz'))
    for i, case in enumerate(cases):
        ref, mov = _pcc_inputs(case, i, arrays)
        s, err, ph = registration.phase_cross_correlation(ref, mov, upsample_factor=10, normalization=None)
        assert s.tolist() == case['shift_none'], (case['shape'], s)
        assert err == pytest.approx(case['error_none'], rel=1e-6, abs=1e-9)
        assert ph == pytest.approx(case['phasediff_none'], abs=1e-9)
        s, _, _ = registration.phase_cross_correlation(ref, mov, upsample_factor=10, normalization='phase')
        assert s.tolist() == case['shift_phase'], (case['shape'], s)
        s, _, _ = registration.phase_cross_correlation(ref, mov, upsample_factor=1, normalization=None)
        assert s.tolist() == case['shift_int'], (case['shape'], s)


@pytest.mark.parametrize('shape', [(64, 64), (32, 128), (50, 36), (45, 64), (64, 45), (33, 35), (256, 80), (128, 2), (2, 64)])
@pytest.mark.parametrize('norm', [None, 'phase'])
def test_random_crops_match_oracle(shape, norm):
    n0, n1 = shape
    rng = np.random.default_rng(n0 * 1000 + n1)
    for trial in range(3):
        dy, dx = int(rng.integers(-n0 // 4, n0 // 4 + 1)), int(rng.integers(-n1 // 4, n1 // 4 + 1))
        big = synth.scene_patch(900 + trial, 0, 0, n0 + 2 * n0, n1 + 2 * n1)
        ref = big[n0:2 * n0, n1:2 * n1].astype(np.uint16)
        mov = (big[n0 - dy:2 * n0 - dy, n1 - dx:2 * n1 - dx] + synth.noise_patch(trial, n0, n1, 300)).astype(np.uint16)
        for u in (1, 10, 4):
            want, werr, wph, detail = O.phase_cross_correlation(ref, mov, u, norm)
            got, gerr, gph = registration.phase_cross_correlation(ref, mov, upsample_factor=u, normalization=norm)
            np.testing.assert_array_equal(got, want, err_msg=f'{shape} u={u} planted={dy, dx}')
            if norm is None:
                assert gerr == pytest.approx(werr, rel=1e-6, abs=1e-7)
                assert abs(np.angle(np.exp(1j * (gph - wph)))) < 1e-7


def test_uint8_and_degenerate_inputs():
    rng = np.random.default_rng(3)
    ref = rng.integers(0, 256, (40, 48)).astype(np.uint8)
    mov = np.roll(ref, (3, -5), axis=(0, 1))
    want = O.phase_cross_correlation(ref, mov, 10, 'phase')[0]
    got = registration.phase_cross_correlation(ref, mov, upsample_factor=10, normalization='phase')[0]
    np.testing.assert_array_equal(got, want)
    # constant images: zero cross-power spectrum, argmax falls on index 0 everywhere
    flat = np.full((32, 32), 7, np.uint16)
    zero = np.zeros((32, 32), np.uint16)
    for a, b in ((zero, zero), (flat, zero)):
        want = O.phase_cross_correlation(a, b, 10, 'phase')[0]
        got = registration.phase_cross_correlation(a, b, upsample_factor=10, normalization='phase')[0]
        np.testing.assert_array_equal(got, want)


def test_bad_arguments():
    a = np.zeros((16, 16), np.uint16)
    with pytest.raises(ValueError, match='same shape'):
        registration.phase_cross_correlation(a, np.zeros((16, 17), np.uint16))
    with pytest.raises(ValueError, match='normalization'):
        registration.phase_cross_correlation(a, a, normalization='bogus')
    with pytest.raises(native.NativeError, match='not supported'):     # a crop side is at most 65535 pixels
        registration.phase_cross_correlation(np.zeros((4, 65536), np.uint16), np.zeros((4, 65536), np.uint16))


def test_tile_minmax_and_normalised_crops_via_grid_center():
    """calculate_shifts on a device plane == oracle on the same tiles (integers bit-exact),
    including the full-tile min-max stretch in front of the crop."""
    import torch
    dev = _dev()
    for spec, pattern in ((synth.GridSpec(rows=3, cols=4, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=12), 'Unidirectional'),
                          (synth.GridSpec(rows=3, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, jy=-3, jx=2, seed=13), 'Unidirectional'),
                          (synth.GridSpec(rows=4, cols=3, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=16,
                                          scan_pattern='S-Pattern', rev_ov_x=48, rev_jy=-2), 'S-Pattern'),
                          (synth.GridSpec(rows=2, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=17, dtype='uint8', noise=0), 'Unidirectional')):
        tiles_np = spec.tile_stack()
        tiles = torch.from_numpy(tiles_np).to(dev)
        mm = native.tile_minmax(tiles).cpu().numpy()
        np.testing.assert_array_equal(mm[:, 0], tiles_np.reshape(len(tiles_np), -1).min(1))
        np.testing.assert_array_equal(mm[:, 1], tiles_np.reshape(len(tiles_np), -1).max(1))
        xs = [spec.stage_mm(0, c)[0] for c in range(spec.cols)]
        ys = [spec.stage_mm(r, 0)[1] for r in range(spec.rows)]
        got = registration.register_grid_center(tiles, spec.rows, spec.cols, xs, ys, spec.pixel_size_um,
                                                spec.pixel_binning, normalization=None, scan_pattern=pattern)
        mx, my = O.max_overlaps(xs, ys, spec.tile_w, spec.tile_h, spec.pixel_size_um, spec.pixel_binning)
        ci, ri = (spec.cols - 1) // 2, (spec.rows - 1) // 2
        t = lambda r, c: tiles_np[r * spec.cols + c]
        dt = tiles_np.dtype.type
        assert got.h_shift == O.calculate_horizontal_shift(t(ri, ci), t(ri, ci + 1), mx, dt, None)
        assert got.v_shift == O.calculate_vertical_shift(t(ri, ci), t(ri + 1, ci), my, dt, None)
        if pattern == 'S-Pattern':
            assert got.h_shift_rev == O.calculate_horizontal_shift(t(ri + 1, ci), t(ri + 1, ci + 1), mx, dt, None)
            assert got.h_shift_rev_odd == int(ri % 2 == 0)


def test_config2_crop_shape_1024x256_and_all_pairs_batch():
    """2048^2 tiles, ov 244 -> 1024x256 / 256x1024 crops; every pair of a 2x3 grid in one batch."""
    import torch
    dev = _dev()
    spec = synth.GridSpec(rows=2, cols=3, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=3000)
    tiles_np = spec.tile_stack()
    tiles = torch.from_numpy(tiles_np).to(dev)
    xs = [spec.stage_mm(0, c)[0] for c in range(3)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(2)]
    mx, my = placement.registration_crop_widths(xs, ys, 2048, 2048, spec.pixel_size_um, spec.pixel_binning)
    assert (mx, my) == (256, 256)
    (hp, (h0, h1)), (vp, (v0, v1)) = registration.all_pairs(2, 3, 2048, 2048, mx, my)
    assert (h0, h1, v0, v1) == (1024, 256, 256, 1024) and len(hp) == 4 and len(vp) == 3
    mm = native.tile_minmax(tiles)
    hs, _, _ = registration.register_pairs(tiles, hp, h0, h1, 10, 'phase', mm)
    vs, _, _ = registration.register_pairs(tiles, vp, v0, v1, 10, 'phase', mm)
    for k, p in enumerate(hp):
        a, b = O.overlap_crops_horizontal(O.normalize_image(tiles_np[p['ref_tile']], np.uint16),
                                          O.normalize_image(tiles_np[p['mov_tile']], np.uint16), mx)
        want = O.phase_cross_correlation(a, b, 10, 'phase')[0]
        np.testing.assert_array_equal(hs[k], want)
        assert registration.horizontal_shift_from(hs[k], h1) == (3, -244)
    for k, p in enumerate(vp):
        a, b = O.overlap_crops_vertical(O.normalize_image(tiles_np[p['ref_tile']], np.uint16),
                                        O.normalize_image(tiles_np[p['mov_tile']], np.uint16), my)
        want = O.phase_cross_correlation(a, b, 10, 'phase')[0]
        np.testing.assert_array_equal(vs[k], want)
        assert registration.vertical_shift_from(vs[k], v0) == (-244, -2)


def test_long_non_power_of_two_lines():
    """Any crop length up to 4860, smooth ones up to 9720: 3000 x 3000 sensors give 1500-long crops (mixed radix),
    6244 x 4168 ones 2084 / 3122 (Bluestein through 4200 / 6250 points), 9568 x 6380 ones 4784 / 3190; 4095, 4096,
    2049 = just past a power of two, 3989 prime; radix-7 / 11 / 13 stages (2002 = 2 7 11 13, 1001); lines beyond
    4096 transformed directly (6000, 8192, 9720) and through Bluestein (4859 -> 9720 points) on either axis: same
    shifts as the oracle."""
    rng = np.random.default_rng(4)
    for n0, n1 in ((1500, 48), (48, 1500), (750, 100), (2084, 214), (218, 3122), (4095, 24), (24, 4095), (4096, 30),
                   (30, 2049), (3989, 20), (20, 3989), (521, 521), (2002, 26), (26, 1001), (6000, 20), (20, 6000),
                   (8192, 16), (16, 9720), (4859, 20), (20, 4784), (3190, 18), (4608, 40), (4620, 40)):
        big = synth.scene_patch(77, 0, 0, n0 + 64, n1 + 64)
        dy, dx = 7, -5
        ref = big[32:32 + n0, 32:32 + n1].astype(np.uint16)
        mov = (big[32 - dy:32 - dy + n0, 32 - dx:32 - dx + n1] + synth.noise_patch(3, n0, n1, 200)).astype(np.uint16)
        for norm in (None, 'phase'):
            want = O.phase_cross_correlation(ref, mov, 10, norm)[0]
            got = registration.phase_cross_correlation(ref, mov, upsample_factor=10, normalization=norm)[0]
            np.testing.assert_array_equal(got, want, err_msg=f'{n0}x{n1} {norm}')
            assert np.abs(want - np.array([-dy, -dx])).max() <= 0.15      # the planted shift, to the 0.1-px grid


def test_lines_too_long_for_the_lds():
    """Crop sides whose line (or Bluestein line) does not fit the 160 KB of LDS are transformed in scratch lines of the
    workspace (round 4; rounds 1-3 refused them): a prime just past the limit (4861 -> a Bluestein line of >= 9721 points),
    2 * 11 * 13 * 17, 9728 = 2^9 * 19, smooth lengths transformed directly (10000 = 2^4 5^4, 16384, 12288 = 2^12 * 3), a
    prime sensor side (9733), both axes long at once, several pairs per launch walking over the scratch lines: same
    shifts as the oracle."""
    for n0, n1 in ((16, 4861), (4862, 12), (9728, 8), (10, 10000), (16384, 6), (8, 12288), (9733, 6), (6, 19997), (4861, 4900)):
        big = synth.scene_patch(78, 0, 0, n0 + 64, n1 + 64)
        dy, dx = (3, -5) if min(n0, n1) > 12 else ((0, -5) if n0 < n1 else (3, 0))
        ref = big[32:32 + n0, 32:32 + n1].astype(np.uint16)
        mov = (big[32 - dy:32 - dy + n0, 32 - dx:32 - dx + n1] + synth.noise_patch(3, n0, n1, 200)).astype(np.uint16)
        for norm in ((None, 'phase') if n0 * n1 < 1 << 20 else (None,)):
            want = O.phase_cross_correlation(ref, mov, 10, norm)[0]
            got = registration.phase_cross_correlation(ref, mov, upsample_factor=10, normalization=norm)[0]
            np.testing.assert_array_equal(got, want, err_msg=f'{n0}x{n1} {norm}')
            assert np.abs(want - np.array([-dy, -dx])).max() <= 0.15, f'{n0}x{n1}: {want}'


def test_long_lines_several_pairs_per_launch():
    """Three pairs with different planted shifts through one launch of the long-line kernels (40 x 4861 crops of 64 x 4900
    tiles, crop origins inside the tiles): every pair its own answer, equal to the oracle's on the same crops."""
    import torch
    dev = _dev()
    n0, n1, H, W = 40, 4861, 64, 4900
    big = synth.scene_patch(79, 0, 0, H + 64, W + 64)
    shifts = [(2, -7), (-3, 4), (0, 9)]
    tiles = [big[32:32 + H, 32:32 + W].astype(np.uint16)]
    for k, (dy, dx) in enumerate(shifts):
        tiles.append((big[32 - dy:32 - dy + H, 32 - dx:32 - dx + W] + synth.noise_patch(5 + k, H, W, 150)).astype(np.uint16))
    stack = torch.from_numpy(np.stack(tiles)).to(dev)
    minmax = torch.tensor([[1, 0]] * len(tiles), dtype=torch.int32, device=dev)      # pixels as they are
    pairs = np.zeros(len(shifts), dtype=native.PAIR_DTYPE)
    for k in range(len(shifts)):
        pairs[k] = (0, 1 + k, 10, 20, 10, 20)
    got, _, _ = registration.register_pairs(stack, pairs, n0, n1, 10, None, minmax)
    for k, (dy, dx) in enumerate(shifts):
        want = O.phase_cross_correlation(tiles[0][10:10 + n0, 20:20 + n1], tiles[1 + k][10:10 + n0, 20:20 + n1], 10, None)[0]
        np.testing.assert_array_equal(got[k], want, err_msg=f'pair {k}')
        assert np.abs(want - np.array([-dy, -dx])).max() <= 0.15


def test_out_of_range_pairs_are_flagged_not_read():
    """The pair table is device memory, so the library cannot validate it on the host: a pair whose
    tile index or crop would leave its tile is skipped on the device and flagged (the Python binding
    refuses such pairs earlier; this goes through the C-ABI underneath it)."""
    import ctypes as C
    import torch
    dev = _dev()
    rng = np.random.default_rng(0)
    tiles = torch.from_numpy(rng.integers(0, 65536, (2, 64, 64)).astype(np.uint16)).to(dev)
    mm = native.tile_minmax(tiles)
    pairs = np.array([(0, 1, 0, 0, 0, 0), (0, 5, 0, 0, 0, 0), (0, 1, 40, 0, 0, 0), (0, 1, 0, 0, 0, -1), (1, 0, 16, 16, 16, 16)],
                     dtype=native.PAIR_DTYPE)
    L = native.lib()
    n0 = n1 = 32
    ws = torch.empty(int(L.sq_register_workspace_bytes(len(pairs), n0, n1, 10)), dtype=torch.uint8, device=dev)
    pd = torch.from_numpy(pairs.view(np.uint8).reshape(-1)).to(dev)
    res = torch.zeros(len(pairs) * native.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    a = native._RegisterArgs()
    a.tile_base_dev, a.tile_stride, a.n_tiles, a.tile_h, a.tile_w, a.tile_pitch, a.tile_dtype = tiles.data_ptr(), 64 * 64, 2, 64, 64, 64, native.SQ_U16
    a.minmax_dev, a.pairs_dev, a.n_pairs, a.n0, a.n1 = mm.data_ptr(), pd.data_ptr(), len(pairs), n0, n1
    a.upsample_factor, a.normalization = 10, native.SQ_NORM_PHASE
    a.results_dev, a.workspace_dev, a.workspace_bytes = res.data_ptr(), ws.data_ptr(), ws.numel()
    assert L.sq_register_pairs(C.byref(a), native._stream_ptr()) == 0
    out = res.cpu().numpy().view(native.RESULT_DTYPE)
    bad = np.iinfo(np.int32).min
    assert [tuple(r['coarse']) == (bad, bad) for r in out] == [False, True, True, True, False]


def test_normalize_tiles_equals_reference_golden_and_oracle():
    import torch
    v = np.load(os.path.join(GOLDEN, 'normalize_vectors.npz'))
    for dt in ('uint16', 'uint8'):
        got = native.normalize_tiles(torch.from_numpy(v[f'in_{dt}'][None]).to(_dev()))[0].cpu().numpy()
        np.testing.assert_array_equal(got, v[f'out_{dt}'])
        got = native.normalize_tiles(torch.from_numpy(v[f'const_in_{dt}'][None]).to(_dev()))[0].cpu().numpy()
        np.testing.assert_array_equal(got, v[f'const_out_{dt}'])       # constant tile: 0/0 -> 0
    rng = np.random.default_rng(6)
    stack = rng.integers(100, 60000, size=(5, 37, 61)).astype(np.uint16)
    stack[3] = 777                                   # max == min: 0/0 -> 0 like the reference's cast
    got = native.normalize_tiles(torch.from_numpy(stack).to(_dev())).cpu().numpy()
    for i in range(5):
        np.testing.assert_array_equal(got[i], O.normalize_image(stack[i], np.uint16))


def test_normalisation_quotient_equals_ieee_division_exhaustively():
    """K1 and sq_normalize_tiles compute (pixel - min) / (max - min) as a multiply by the reciprocal of the range plus
    Markstein's correction; on the device, for ALL 65536 x 65536 (numerator, range) pairs of 16-bit integers, the result
    is the IEEE float64 quotient bit for bit (NaN for 0 / 0 either way)."""
    import torch
    assert native.selftest_normalise_divide(torch.device('cuda:0')) == 0

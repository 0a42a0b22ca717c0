#!/opt/conda/bin/python3.9
"""Generate golden vectors by executing the REAL reference hot path.

Run in the authoring container only (the reference never travels):

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

What it does: registers inert stand-ins for the six I/O-only packages that are
absent offline (cv2, dask_image, zarr, ome_zarr, aicsimageio, basicpy: none of
them is on the arithmetic path), imports the unmodified
``/root/reference/stitcher.py``, builds seeded synthetic acquisitions with this
repo's generator (image-stitcher_amd/synth.py), runs the reference's own methods
(parse_acquisition_metadata, calculate_shifts, stitch_region, ...) and dumps
inputs' specs + outputs as small ``.npz``/``.json`` fixtures next to this file.

skimage here is 0.18.3 (no ``normalization`` argument).  ``phase`` mode vectors are
produced by feeding the phase-normalised cross-power spectrum through the real
0.18.3 routine with ``space='fourier'`` (peak search and upsampled refinement stay
third-party code) -- see SURVEY.md 8c.
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)


def _install_standins():
    import dask.array as da
    import tifffile

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('cv2', imwrite=lambda *a, **k: True)
    di = mod('dask_image')
    di.imread = mod('dask_image.imread',
                    imread=lambda p: da.from_array(tifffile.imread(p)[None], chunks=-1))
    mod('zarr')
    mod('ome_zarr')
    aw = mod('aicsimageio.writers', OmeTiffWriter=object, OmeZarrWriter=object)
    at = mod('aicsimageio.types')
    mod('aicsimageio', writers=aw, types=at)
    mod('basicpy', BaSiC=object)


_install_standins()
sys.path.insert(0, REF)
import stitcher as ref_stitcher                      # noqa: E402  (the reference, unmodified)
from stitcher_parameters import StitchingParameters  # noqa: E402
from skimage.registration import phase_cross_correlation as sk_pcc  # noqa: E402

import image_stitcher_amd.synth as synth             # noqa: E402


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


class Recorder:
    """Wraps the reference module's pcc symbol and one Stitcher's tile placement."""

    def __init__(self):
        self.pcc_calls = []
        self.placements = []

    def pcc(self, a, b, **kw):
        a_np, b_np = np.asarray(a), np.asarray(b)
        shift, err, phase = sk_pcc(a_np, b_np, **kw)
        self.pcc_calls.append(dict(ref=a_np.copy(), mov=b_np.copy(), shift=np.array(shift, dtype=np.float64),
                                   error=float(err), phasediff=float(phase)))
        return shift, err, phase


def run_case(name, spec, *, use_registration, apply_flatfield=False, flat_dtype=None,
             registration_channel='', registration_z_level=0, keep_canvas=True, windows=(), forced=None):
    tmp = tempfile.mkdtemp(prefix='golden_')
    root = os.path.join(tmp, 'acq')
    try:
        synth.write_acquisition(spec, root)
        params = StitchingParameters(input_folder=root, use_registration=use_registration,
                                     apply_flatfield=apply_flatfield,
                                     registration_channel=registration_channel,
                                     registration_z_level=registration_z_level,
                                     scan_pattern=spec.scan_pattern)
        st = ref_stitcher.Stitcher(params)
        rec = Recorder()
        ref_stitcher.phase_cross_correlation = rec.pcc
        st.get_timepoints()
        st.extract_acquisition_parameters()
        st.get_pixel_size()
        st.parse_acquisition_metadata()
        out = dict(name=name)
        arrays = {}
        if apply_flatfield:
            for ci in range(st.num_c):
                ff = synth.synthetic_flatfield(spec.tile_h, spec.tile_w, np.dtype(flat_dtype))
                # make channels differ a little, deterministically
                st.flatfields[ci] = (ff * np.dtype(flat_dtype).type(1.0 + 0.03125 * ci)).astype(flat_dtype)
        if use_registration:
            if forced is None:
                st.calculate_shifts(st.timepoints[0], st.regions[0])
            else:   # shifts dictated, not measured: the reference's geometry code on arbitrary integers
                st.h_shift, st.v_shift = tuple(forced[0]), tuple(forced[1])
                if len(forced) > 2:
                    st.h_shift_rev, st.h_shift_rev_odd = tuple(forced[2]), forced[3]
                st.calculate_output_dimensions(st.timepoints[0], st.regions[0])   # sets x/y_positions as calculate_shifts would
                out['forced'] = True
            out['h_shift'] = [int(v) for v in st.h_shift]
            out['v_shift'] = [int(v) for v in st.v_shift]
            if spec.scan_pattern == 'S-Pattern':
                out['h_shift_rev'] = [int(v) for v in st.h_shift_rev]
                out['h_shift_rev_odd'] = int(st.h_shift_rev_odd)
            for i, c in enumerate(rec.pcc_calls):
                out.setdefault('pcc', []).append(dict(
                    shape=list(c['ref'].shape), shift=[float(v) for v in c['shift']],
                    error=c['error'], phasediff=c['phasediff'],
                    ref_sha=sha(c['ref']), mov_sha=sha(c['mov'])))
                if c['ref'].size <= 64 * 1024:
                    arrays[f'pcc{i}_ref'] = c['ref']
                    arrays[f'pcc{i}_mov'] = c['mov']
        out['regions'] = list(st.regions)
        out['timepoints'] = [int(t) for t in st.timepoints]
        out['channels'] = list(st.monochrome_channels)
        out['num_z'] = int(st.num_z)
        out['dtype'] = str(np.dtype(st.dtype))
        out['canvases'] = {}
        orig_place = st.place_single_channel_tile
        for t in st.timepoints:
            for region in st.regions:
                placements = []

                def wrapped(stitched_region, tile, x_pixel, y_pixel, z_level, channel_idx, tt,
                            _pl=placements):
                    # record what the reference is about to do, then let it do it
                    _pl.append([int(channel_idx), int(z_level), int(x_pixel), int(y_pixel),
                                int(getattr(st, 'row_index', -1)), int(getattr(st, 'col_index', -1))])
                    return orig_place(stitched_region, tile, x_pixel, y_pixel, z_level, channel_idx, tt)

                st.place_single_channel_tile = wrapped
                canvas = st.stitch_region(int(t), region)
                canvas = np.asarray(canvas.compute() if hasattr(canvas, 'compute') else canvas)
                key = f't{int(t)}_{region}'
                info = dict(shape=list(canvas.shape), sha256=sha(canvas),
                            num_pyramid_levels=int(st.num_pyramid_levels),
                            n_placements=len(placements))
                # (channel, z, x_pixel, y_pixel, row, col) per file, in the reference's write order
                arrays[f'{key}_placements'] = np.array(placements, dtype=np.int64)
                if keep_canvas:
                    arrays[f'{key}_canvas'] = canvas
                for wi, (c, z, y0, x0, hh, ww) in enumerate(windows):
                    arrays[f'{key}_win{wi}'] = canvas[0, c, z, y0:y0 + hh, x0:x0 + ww].copy()
                    info.setdefault('windows', []).append([c, z, y0, x0, hh, ww])
                out['canvases'][key] = info
        out['spec'] = {k: (list(v) if isinstance(v, tuple) else v)
                       for k, v in synth.dataclasses.asdict(spec).items()}
        out['params'] = dict(use_registration=use_registration, apply_flatfield=apply_flatfield,
                             flat_dtype=str(np.dtype(flat_dtype)) if flat_dtype else None,
                             registration_channel=registration_channel,
                             registration_z_level=registration_z_level)
        with open(os.path.join(HERE, f'{name}.json'), 'w') as fh:
            json.dump(out, fh, indent=1)
        if arrays:
            np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **arrays)
        print(f'[golden] {name}: h={out.get("h_shift")} v={out.get("v_shift")} '
              f'canvases={ {k: v["shape"] for k, v in out["canvases"].items()} }')
        return out
    finally:
        ref_stitcher.phase_cross_correlation = sk_pcc
        shutil.rmtree(tmp, ignore_errors=True)


def pcc_vectors():
    """Raw skimage-0.18.3 phase_cross_correlation outputs on seeded crops, plus 'phase'
    mode through the real routine in fourier space."""
    cases = []
    arrays = {}
    eps = np.finfo(np.float64).eps
    shapes = [(64, 34), (34, 64), (256, 80), (80, 256), (96, 50), (45, 64), (1024, 256), (256, 1024),
              (63, 33), (50, 47), (64, 34), (33, 128), (128, 127),   # odd widths; shifts near half the crop (wrap-around)
              # lengths with large prime factors (a 6244 x 4168 sensor: crops 2084 = 4 * 521 and 3122 = 2 * 7 * 223
              # long), a smooth one (1500), 2 * 107, primes (1031, 521), and just past a power of two (2049)
              (2084, 214), (214, 3122), (1500, 107), (1031, 80), (96, 521), (48, 2049),
              # round 3: lines beyond 4096 -- smooth lengths transformed directly by mixed radix (6000 = 2^4 3 5^3, 300,
              # 4410 = 2 3^2 5 7^2, 9720 = 2^3 3^5 5 = the longest line at all), a 9568 x 6380 sensor's crops
              # (4784 = 2^4 13 23 and 3190 = 2 5 11 29: Bluestein through 9600 and 6400 points), and the longest
              # Bluestein lines (4859 = 43 * 113 and 4858 = 2 * 7 * 347, both through 9720 points)
              (6000, 300), (300, 6000), (4784, 60), (60, 3190), (4410, 52), (40, 9720), (4859, 34), (44, 4858),
              # round 4: lines that do not fit the LDS (transformed in the workspace): a prime just past the limit (4861, Bluestein
              # through >= 9721 points), 4862 = 2 11 13 17, a prime sensor side (9733), smooth lengths transformed directly
              # (10000 = 2^4 5^4, 16384), 9728 = 2^9 19
              (16, 4861), (4862, 12), (9733, 6), (10, 10000), (16384, 6), (9728, 8)]
    for i, (n0, n1) in enumerate(shapes):
        seed = 4242 + i
        dy, dx = [(3, -2), (-4, 5), (0, 0), (7, 1), (-1, -6), (2, 2), (5, -3), (-2, 4),
                  (4, -3), (-5, 6), (12, -11), (-9, 14), (1, -13),
                  (6, -9), (-7, 11), (13, 3), (-15, -4), (9, 16), (-3, -12),
                  (11, -6), (-8, 14), (5, 9), (-12, -7), (15, -2), (-4, 10), (7, -15), (-10, 3),
                  (3, -14), (-11, 2), (9, 1), (-2, 13), (-14, -1), (6, 2)][i]
        big = synth.scene_patch(seed, 100, 100, n0 + 32, n1 + 32)      # 16-px margin: |planted| <= 16
        ref = big[16:16 + n0, 16:16 + n1]
        mov = big[16 - dy:16 - dy + n0, 16 - dx:16 - dx + n1] + synth.noise_patch(seed + 1, n0, n1, 150)
        # min-max stretch as uint16, like the crops the stitcher hands to skimage
        def stretch(a):
            a = a.astype(np.uint16)
            return ((a - a.min()) / (a.max() - a.min()) * 65535).astype(np.uint16)
        ref, mov = stretch(ref), stretch(mov)
        s_none, e_none, p_none = sk_pcc(ref, mov, upsample_factor=10)
        F = np.fft.fftn(ref)
        G = np.fft.fftn(mov)
        P = F * G.conj()
        Pn = P / np.maximum(np.abs(P), 100 * eps)
        s_phase, _, _ = sk_pcc(Pn, np.ones_like(Pn), space='fourier', upsample_factor=10)
        s_int, _, _ = sk_pcc(ref, mov, upsample_factor=1)
        cases.append(dict(shape=[n0, n1], seed=seed, planted=[dy, dx],
                          shift_none=[float(v) for v in s_none], error_none=float(e_none),
                          phasediff_none=float(p_none),
                          shift_phase=[float(v) for v in s_phase],
                          shift_int=[float(v) for v in s_int],
                          ref_sha=sha(ref), mov_sha=sha(mov)))
        if n0 * n1 <= 64 * 1024:
            arrays[f'ref{i}'] = ref
            arrays[f'mov{i}'] = mov
        print(f'[golden] pcc {n0}x{n1}: planted={dy, dx} none={s_none} phase={s_phase}')
    with open(os.path.join(HERE, 'pcc_vectors.json'), 'w') as fh:
        json.dump(cases, fh, indent=1)
    np.savez_compressed(os.path.join(HERE, 'pcc_vectors.npz'), **arrays)


def flatfield_vectors():
    """apply_flatfield_correction (stitcher.py:607-611) on one tile, float32 and float64
    flatfields, including zero / tiny / huge gains to hit the clip and the inf path."""
    spec = synth.GridSpec(rows=1, cols=1, tile_h=48, tile_w=64, ov_y=0, ov_x=0, seed=77)
    tile = spec.tile(0, 0)
    tile[0, :8] = [0, 1, 65535, 65534, 32768, 2, 3, 40000]
    arrays = dict(tile=tile)

    class Dummy:
        dtype = np.uint16
    d = Dummy()
    for dt in (np.float32, np.float64):
        ff = synth.synthetic_flatfield(48, 64, dt).copy()
        ff[0, :8] = np.array([0.0, 0.0, 0.5, 1.0000001, 0.3, 1e-9, 3.0, 0.61], dtype=dt)
        ff[1, :4] = np.array([0.1, 0.7, 0.9999999, 1.5], dtype=dt)
        d.flatfields = {0: ff}
        with np.errstate(all='ignore'):
            out = ref_stitcher.Stitcher.apply_flatfield_correction(d, tile, 0)
        arrays[f'ff_{np.dtype(dt).name}'] = ff
        arrays[f'out_{np.dtype(dt).name}'] = np.asarray(out)
    # channel without a flatfield: identity
    d.flatfields = {}
    assert ref_stitcher.Stitcher.apply_flatfield_correction(d, tile, 0) is tile
    np.savez_compressed(os.path.join(HERE, 'flatfield_vectors.npz'), **arrays)
    print('[golden] flatfield vectors written')


def normalize_vectors():
    """normalize_image (stitcher.py:613-617) on uint16 and uint8 tiles."""
    arrays = {}

    class Dummy:
        pass
    for dt in ('uint16', 'uint8'):
        spec = synth.GridSpec(rows=1, cols=1, tile_h=40, tile_w=56, ov_y=0, ov_x=0, seed=5, dtype=dt)
        tile = spec.tile(0, 0)
        d = Dummy()
        d.dtype = np.dtype(dt).type
        arrays[f'in_{dt}'] = tile
        arrays[f'out_{dt}'] = np.asarray(ref_stitcher.Stitcher.normalize_image(d, tile))
        # a constant tile: (img - min) / (max - min) is 0/0 everywhere; the cast of NaN decides the result
        const = np.full((8, 12), 1234 % (np.iinfo(dt).max + 1), dtype=dt)
        with np.errstate(all='ignore'):
            arrays[f'const_in_{dt}'] = const
            arrays[f'const_out_{dt}'] = np.asarray(ref_stitcher.Stitcher.normalize_image(d, const))
    np.savez_compressed(os.path.join(HERE, 'normalize_vectors.npz'), **arrays)
    print('[golden] normalize vectors written')


def pyramid_vectors():
    """What ome_zarr's Scaler.nearest computes per plane and level (stitcher.py:797-798; ome_zarr itself is
    absent offline): skimage.transform.resize(plane, (Y//2, X//2), order=0, preserve_range=True,
    anti_aliasing=False).  Two third-party executions are recorded:
      * ``sk_*``: that call through scikit-image 0.18.3 (here), for sizes with an odd dimension.  For an
        exactly halved (even) dimension 0.18.3's warp lands on the tie 2*o + 0.5 and resolves it by
        rounding noise (neither [0::2] nor [1::2]); the recorded ``sk_even_*`` vector shows that, it is not
        a parity target;
      * ``zoom_*``: scipy.ndimage.zoom(plane, out/in, order=0, mode='mirror', grid_mode=True), the code
        path scikit-image >= 0.19 takes for the same resize call; even and odd sizes."""
    from scipy import ndimage as ndi
    from skimage.transform import resize
    rng = np.random.default_rng(77)
    arrays = {}
    for i, (shape, dt) in enumerate([((37, 91), 'uint16'), ((513, 255), 'uint16'), ((5, 3), 'uint8'),
                                     ((101, 77), 'uint8'), ((64, 96), 'uint16'), ((130, 57), 'uint16'),
                                     ((2, 2), 'uint16'), ((3, 2), 'uint8')]):
        a = rng.integers(0, np.iinfo(dt).max + 1, shape).astype(dt)
        out_shape = (shape[0] // 2, shape[1] // 2)
        arrays[f'in_{i}'] = a
        sk = resize(a, output_shape=out_shape, order=0, preserve_range=True, anti_aliasing=False).astype(a.dtype)
        both_odd_or_tiny = all(n % 2 == 1 for n in shape)
        arrays[('sk_' if both_odd_or_tiny else 'sk_even_') + str(i)] = sk
        arrays[f'zoom_{i}'] = ndi.zoom(a, [o / n for o, n in zip(out_shape, shape)], order=0, mode='mirror',
                                       grid_mode=True)
        assert arrays[f'zoom_{i}'].shape == out_shape
    np.savez_compressed(os.path.join(HERE, 'pyramid_vectors.npz'), **arrays)
    print('[golden] pyramid vectors written')


def degenerate_cases():
    """One row, one column, one tile (coordinate-only; with -r the reference dies with IndexError at
    stitcher.py:444-445, which tests/test_host_cpu.py expects of the drop-in too)."""
    G = synth.GridSpec
    for name, rows, cols in (('coord_1x3', 1, 3), ('coord_3x1', 3, 1), ('coord_1x1', 1, 1)):
        run_case(name, G(rows=rows, cols=cols, tile_h=64, tile_w=96, ov_y=16, ov_x=24, seed=21, nz=2), use_registration=False)


def more_cases():
    """Added late in round 1: uint8 tiles with gains (clip at 255), S-Pattern with an even centre row,
    a registration channel that does not exist (fallback to the first channel, stitcher.py:433-437), stage
    positions off the grid in coordinate mode (every FOV has its own x and y: the int() truncations)."""
    G = synth.GridSpec
    ch2 = synth.DEFAULT_CHANNELS[:2]
    run_case('reg_uint8_ff32', G(rows=2, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=23, dtype='uint8', noise=0,
                                 channels=ch2),
             use_registration=True, apply_flatfield=True, flat_dtype='float32')
    run_case('reg_spattern_even', G(rows=5, cols=3, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=24,
                                    scan_pattern='S-Pattern', rev_ov_x=48, rev_jy=-2),
             use_registration=True)
    run_case('reg_missing_channel', G(rows=3, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=25, channels=ch2),
             use_registration=True, registration_channel='Fluorescence 730 nm Ex')
    run_case('reg_blank_centre', G(rows=3, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=27, blank_fovs=(4,)),
             use_registration=True)
    # pixel_binning 1: the crop width formula halves the overlap (odd width 21) and the shifts are whatever
    # scikit-image makes of it -- deterministic all the same
    run_case('reg_binning1', G(rows=3, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=28, pixel_binning=1),
             use_registration=True)
    run_case('reg_binning1_odd', G(rows=3, cols=3, tile_h=127, tile_w=131, ov_y=37, ov_x=43, seed=29, pixel_binning=1),
             use_registration=True)
    # a plate whose wells hold different grids (edge wells with fewer FOVs): shifts measured on the first well
    run_case('reg_unequal_wells', G(rows=3, cols=3, tile_h=96, tile_w=128, ov_y=24, ov_x=40, seed=30, regions=('A1', 'A2', 'B1'),
                                    region_dims=(('A2', 2, 3), ('B1', 3, 2)), channels=ch2),
             use_registration=True)
    run_case('coord_unequal_wells', G(rows=3, cols=3, tile_h=96, tile_w=128, ov_y=24, ov_x=40, seed=31, regions=('A1', 'B2'),
                                      region_dims=(('B2', 1, 2),), nt=2),
             use_registration=False)
    run_case('coord_jitter', G(rows=3, cols=4, tile_h=96, tile_w=128, ov_y=24, ov_x=40, seed=26, nz=2, stage_jitter_um=3.7),
             use_registration=False)


def long_line_cases():
    """Round 4: a sensor whose registration crops do not fit the LDS as FFT lines -- 96 x 19466 tiles: the vertical pair's crop is
    (overlap) x 9734 = 2 * 4867 (a Bluestein line of >= 19467 points, transformed in the workspace by the device path); the
    reference (pocketfft) takes any length (stitcher.py:503-510, 516-523)."""
    G = synth.GridSpec
    run_case('reg_long_lines', G(rows=2, cols=2, tile_h=96, tile_w=19466, ov_y=28, ov_x=310, seed=3100),
             use_registration=True, keep_canvas=False,
             windows=[(0, 0, 60, 19100, 48, 96), (0, 0, 0, 0, 16, 128), (0, 0, 120, 38000, 40, 80)])


def forced_cases():
    """The reference's integer geometry (canvas size, placement, crops, clip: stitcher.py:298-354, 563-605,
    639-689) driven with dictated shifts of every sign combination -- including ones no registration of real
    tiles would return (positive 'overlaps', large skews): floor division on negatives, abs() here but not
    there, int() truncation."""
    G = synth.GridSpec
    combos = [((5, -20), (-20, -3)), ((-4, -18), (-22, 6)), ((0, 10), (12, 0)), ((7, 0), (0, -7)), ((-9, 15), (-14, -11)),
              ((11, -31), (9, 13)), ((-1, -1), (-1, -1)), ((0, 0), (0, 0)), ((3, -79), (-63, -2))]
    for k, (h, v) in enumerate(combos):
        try:
            run_case(f'forced_{k}', G(rows=3, cols=4, tile_h=64, tile_w=80, ov_y=16, ov_x=20, seed=40 + k, nz=1),
                     use_registration=True, forced=(h, v))
        except Exception as exc:      # the reference itself cannot place tiles with these shifts
            print(f'[golden] forced_{k} {h} {v}: reference raises {type(exc).__name__}: {exc}')
    run_case('forced_spattern', G(rows=4, cols=3, tile_h=64, tile_w=80, ov_y=16, ov_x=20, seed=60, scan_pattern='S-Pattern'),
             use_registration=True, forced=((4, -22), (-17, 3), (-3, -25), True))


def main():
    if sys.argv[1:] == ['forced']:
        return forced_cases()
    if sys.argv[1:] == ['long']:
        return long_line_cases()
    if sys.argv[1:] == ['more']:
        return more_cases()
    if sys.argv[1:] == ['normalize']:
        return normalize_vectors()
    if sys.argv[1:] == ['pcc']:
        return pcc_vectors()
    if sys.argv[1:] == ['pyramid']:
        return pyramid_vectors()
    if sys.argv[1:] == ['degenerate']:
        return degenerate_cases()
    G = synth.GridSpec
    ch2 = synth.DEFAULT_CHANNELS[:2]
    # config 1: 2x2 of 512^2, coordinate-only (BASELINE.json configs[0])
    run_case('cfg1_coord_2x2_512', G(rows=2, cols=2, tile_h=512, tile_w=512, ov_y=77, ov_x=77, seed=1000),
             use_registration=False, keep_canvas=False,
             windows=[(0, 0, 400, 400, 96, 96), (0, 0, 0, 0, 32, 64), (0, 0, 880, 860, 67, 87)])
    # >= 11 FOVs so '_10_' sorts before '_2_': last-writer-wins order
    run_case('coord_3x4_small', G(rows=3, cols=4, tile_h=96, tile_w=128, ov_y=24, ov_x=40, seed=11,
                                  channels=ch2, nz=2),
             use_registration=False)
    run_case('reg_3x4_small', G(rows=3, cols=4, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=12,
                                channels=ch2, nz=2),
             use_registration=True, registration_channel=ch2[1], registration_z_level=1)
    run_case('reg_neg_skew', G(rows=3, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, jy=-3, jx=2, seed=13),
             use_registration=True)
    run_case('reg_ff32', G(rows=3, cols=3, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=14, channels=ch2),
             use_registration=True, apply_flatfield=True, flat_dtype='float32')
    run_case('reg_ff64', G(rows=3, cols=3, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=14, channels=ch2),
             use_registration=True, apply_flatfield=True, flat_dtype='float64')
    run_case('coord_ff32', G(rows=2, cols=3, tile_h=96, tile_w=96, ov_y=20, ov_x=20, seed=15),
             use_registration=False, apply_flatfield=True, flat_dtype='float32')
    run_case('reg_spattern', G(rows=4, cols=3, tile_h=128, tile_w=160, ov_y=36, ov_x=44, seed=16,
                               scan_pattern='S-Pattern', rev_ov_x=48, rev_jy=-2),
             use_registration=True)
    run_case('reg_uint8', G(rows=2, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=17, dtype='uint8',
                            noise=0),
             use_registration=True)
    run_case('reg_multi', G(rows=2, cols=2, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=18,
                            regions=('A1', 'B2'), nt=2),
             use_registration=True)
    run_case('reg_ragged', G(rows=3, cols=3, tile_h=128, tile_w=128, ov_y=40, ov_x=40, seed=19, channels=ch2, nz=2,
                             missing=((6, 0, 1, 0), (8, 1, 1, 0), (2, 1, 0, 0))),
             use_registration=True)
    run_case('coord_rgb', G(rows=2, cols=2, tile_h=64, tile_w=96, ov_y=16, ov_x=24, seed=20, dtype='uint8',
                            channels=('BF LED matrix full_RGB', 'Fluorescence 488 nm Ex'),
                            rgb_channels=('BF LED matrix full_RGB',)),
             use_registration=False)
    run_case('reg_3x3_512', G(rows=3, cols=3, tile_h=512, tile_w=512, ov_y=77, ov_x=77, seed=2000),
             use_registration=True, keep_canvas=False,
             windows=[(0, 0, 420, 420, 64, 64), (0, 0, 0, 0, 16, 128), (0, 0, 860, 880, 80, 80)])
    run_case('reg_2x2_2048', G(rows=2, cols=2, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=3000),
             use_registration=True, keep_canvas=False,
             windows=[(0, 0, 1780, 1780, 64, 64), (0, 0, 3600, 3500, 48, 96)])
    pcc_vectors()
    flatfield_vectors()
    normalize_vectors()
    pyramid_vectors()
    degenerate_cases()
    more_cases()
    forced_cases()


if __name__ == '__main__':
    main()

"""Parity of the HIP fusion kernels (through the C-ABI) against the oracle.  Bit-exact for
overwrite mode (uint16/uint8, with and without flatfield); feather mode is an extension and
is held to 1e-5 relative against the oracle's definition."""
import numpy as np
import pytest

from helpers import REGION_CASES  # noqa: F401  (keeps helpers importable on the GPU box)
from image_stitcher_amd import native, synth
from oracle import stitch_oracle as O

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch


def run_fuse(rects, tiles_np, ch, cw, mode=native.SQ_FUSE_OVERWRITE, flats_np=None, out_dtype=None, use_ptrs=False,
             n_planes=1, flags=0):
    torch = _torch()
    dev = torch.device('cuda:0')
    th, tw = tiles_np.shape[-2:]
    plan = native.FusePlan(rects, th, tw, ch, cw, mode)
    tiles = torch.from_numpy(np.ascontiguousarray(tiles_np)).to(dev)
    out_dtype = out_dtype or tiles_np.dtype
    canvas = torch.full((n_planes, ch, cw), 7, dtype=native.torch_dtype_of(out_dtype), device=dev)
    flats = None
    if flats_np is not None:
        flats = [None if f is None else torch.from_numpy(f).to(dev) for f in flats_np]
    if use_ptrs:
        flat_list = tiles.reshape(-1, th, tw)
        ptrs = native.pointer_table([flat_list[i] for i in range(flat_list.shape[0])], dev)
        native.fuse_planes(plan, None if mode == native.SQ_FUSE_OVERWRITE else tiles, canvas, flats, tile_ptrs=ptrs,
                           flags=flags)
    else:
        native.fuse_planes(plan, tiles, canvas, flats, flags=flags)
    torch.cuda.synchronize()
    return canvas.cpu().numpy(), plan


def random_rects(rng, n, th, tw, ch, cw):
    rects = []
    for _ in range(n):
        sy, sx = int(rng.integers(0, th // 3)), int(rng.integers(0, tw // 3))
        rects.append((sy, sx, int(rng.integers(1, th - sy + 1)), int(rng.integers(1, tw - sx + 1)),
                      int(rng.integers(0, ch)), int(rng.integers(0, cw))))
    return np.array(rects)


@pytest.mark.parametrize('dtype', ['uint16', 'uint8'])
@pytest.mark.parametrize('seed', range(4))
def test_overwrite_random_rects(seed, dtype):
    rng = np.random.default_rng(seed)
    th, tw = int(rng.integers(30, 90)), int(rng.integers(30, 200))
    ch, cw = int(rng.integers(50, 400)), int(rng.integers(50, 700))    # odd pitches: every alignment phase
    n = int(rng.integers(1, 30))
    rects = random_rects(rng, n, th, tw, ch, cw)
    tiles = rng.integers(0, np.iinfo(dtype).max + 1, size=(1, n, th, tw)).astype(dtype)
    got, _ = run_fuse(rects, tiles, ch, cw, use_ptrs=bool(seed % 2))
    want = O.fuse_plane_overwrite(list(tiles[0]), rects, ch, cw)
    np.testing.assert_array_equal(got[0], want)


@pytest.mark.parametrize('fdtype', ['float32', 'float64'])
@pytest.mark.parametrize('dtype', ['uint16', 'uint8'])
def test_overwrite_flatfield_multi_plane(dtype, fdtype):
    rng = np.random.default_rng(11)
    th, tw, ch, cw, n, planes = 64, 100, 231, 333, 9, 3
    rects = random_rects(rng, n, th, tw, ch - 20, cw - 20)
    tiles = rng.integers(0, np.iinfo(dtype).max + 1, size=(planes, n, th, tw)).astype(dtype)
    flats = []
    for p in range(planes):
        f = (0.3 + 1.4 * rng.random((th, tw))).astype(fdtype)
        f[0, :6] = [0.0, 1e-9, 1.0, 0.5, 3.0, 1.0000001]     # inf / clip / exact paths
        flats.append(f)
    flats[1] = None      # a channel without a flatfield: identity (stitcher.py:608,611)
    tiles[:, :, 0, 0] = 0   # 0/0 -> NaN -> 0
    got, _ = run_fuse(rects, tiles, ch, cw, flats_np=flats, n_planes=planes)
    for p in range(planes):
        want = O.fuse_plane_overwrite(list(tiles[p]), rects, ch, cw, flats[p])
        np.testing.assert_array_equal(got[p], want)


def test_flatfield_golden_vectors_through_kernel():
    import os
    from helpers import GOLDEN
    v = np.load(os.path.join(GOLDEN, 'flatfield_vectors.npz'))
    tile = v['tile']
    th, tw = tile.shape
    rects = np.array([(0, 0, th, tw, 0, 0)])
    for dt in ('float32', 'float64'):
        got, _ = run_fuse(rects, tile[None, None], th, tw, flats_np=[v[f'ff_{dt}']])
        np.testing.assert_array_equal(got[0], v[f'out_{dt}'])


def test_grid_geometry_like_reference_config2_scaled():
    """Registered 8x8 grid geometry (crops, skew, oversize canvas) at 1/8 scale."""
    spec = synth.GridSpec(rows=8, cols=8, tile_h=256, tile_w=256, ov_y=30, ov_x=30, seed=2)
    h_shift, v_shift = (3, -30), (-30, -2)
    xs = [spec.stage_mm(0, c)[0] for c in range(8)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(8)]
    wc, hc, _ = O.output_dimensions(xs, ys, 256, 256, spec.pixel_size_um, True, h_shift, v_shift)
    rects, tiles = [], []
    order = sorted(range(64), key=lambda f: str(f))        # sorted-filename order of fov numbers
    for fov in order:
        r, c = divmod(fov, 8)
        info = dict(x=xs[c], y=ys[r])
        x_px, y_px, top, bottom, left, right = O.tile_rect(info, xs, ys, 256, 256, spec.pixel_size_um, True, h_shift,
                                                           v_shift, None, 0, wc, hc)
        rects.append((top, left, 256 - top - bottom, 256 - left - right, y_px + top, x_px + left))
        tiles.append(spec.tile(r, c))
    rects = np.array(rects)
    tiles = np.stack(tiles)[None]
    got, plan = run_fuse(rects, tiles, hc, wc)
    want = O.fuse_plane_overwrite(list(tiles[0]), rects, hc, wc)
    np.testing.assert_array_equal(got[0], want)
    assert plan.covered_voxels == int((want != 0).sum())


@pytest.mark.parametrize('out_dtype', ['float32', 'uint16'])
@pytest.mark.parametrize('with_flat', [False, True])
def test_feather_matches_oracle_definition(out_dtype, with_flat):
    rng = np.random.default_rng(21)
    th, tw, ch, cw = 48, 64, 150, 190
    rects = np.array([(0, 0, th, tw, int(y), int(x)) for y, x in
                      zip(rng.integers(0, ch - 30, 10), rng.integers(0, cw - 30, 10))])
    tiles = rng.integers(0, 65536, size=(1, 10, th, tw)).astype(np.uint16)
    flat = (0.6 + 0.8 * rng.random((th, tw))).astype(np.float32) if with_flat else None
    got, plan = run_fuse(rects, tiles, ch, cw, mode=native.SQ_FUSE_FEATHER, flats_np=[flat] if with_flat else None,
                         out_dtype=out_dtype)
    want = O.fuse_plane_feather(list(tiles[0]), rects, ch, cw, flat, out_dtype=np.dtype(out_dtype).type)
    assert plan.max_refs >= 2
    # north_star asks for 1e-5 relative; the kernel keeps numpy's operation order, so it is exact
    np.testing.assert_allclose(got[0], want, rtol=1e-5, atol=0)
    np.testing.assert_array_equal(got[0], want)


def test_synth_device_generator_equals_numpy():
    torch = _torch()
    for dtype in ('uint16', 'uint8'):
        spec = synth.GridSpec(rows=2, cols=3, tile_h=40, tile_w=72, ov_y=10, ov_x=12, seed=31, dtype=dtype,
                              jy=-3, jx=2, base=5)   # base 5 with negative drift: negative scene coordinates
        desc = np.zeros(6, dtype=native.SYNTH_DTYPE)
        for r in range(2):
            for c in range(3):
                i = r * 3 + c
                oy, ox = spec.origin(r, c)
                desc[i] = (spec.scene_seed(0, 0, 1, 2) & (2 ** 64 - 1),
                           spec.noise_seed(0, 0, 1, 2, spec.fov_index(r, c)) & (2 ** 64 - 1), oy - 20, ox - 20)
        got = native.synth_tiles(desc, 40, 72, spec.noise, dtype, torch.device('cuda:0')).cpu().numpy()
        for r in range(2):
            for c in range(3):
                oy, ox = spec.origin(r, c)
                v = synth.scene_patch(spec.scene_seed(0, 0, 1, 2), oy - 20, ox - 20, 40, 72) + \
                    synth.noise_patch(spec.noise_seed(0, 0, 1, 2, spec.fov_index(r, c)), 40, 72, spec.noise)
                want = (v >> 8).astype(np.uint8) if dtype == 'uint8' else v.astype(np.uint16)
                np.testing.assert_array_equal(got[r * 3 + c], want)


def test_fuse_argument_errors_are_reported():
    torch = _torch()
    dev = torch.device('cuda:0')
    plan = native.FusePlan(np.array([(0, 0, 8, 8, 0, 0)]), 8, 8, 16, 16)
    tiles = torch.zeros((1, 1, 8, 8), dtype=torch.uint16, device=dev)
    with pytest.raises(native.NativeError, match='overwrite mode keeps the tile dtype'):
        native.fuse_planes(plan, tiles, torch.zeros((1, 16, 16), dtype=torch.float32, device=dev))
    with pytest.raises(native.NativeError, match='geometry differs'):
        native.fuse_planes(plan, tiles, torch.zeros((1, 16, 24), dtype=torch.uint16, device=dev))
    with pytest.raises(ValueError, match='elements'):
        native.fuse_planes(plan, tiles[:, :, :4], torch.zeros((1, 16, 16), dtype=torch.uint16, device=dev))


def test_fast_flatfield_divide_is_exact_exhaustively():
    """The shortened float32 divide of the fusion kernels gives the same truncated (overwrite mode) and
    the same rounded (feather mode) clipped integer as the IEEE quotient for every
    gain it is allowed to see (all 2^23 mantissas of all 200 binades 2^-100..2^99) x every uint16
    numerator: 1.1e14 pairs, scalar and pair forms; and for negative gains in a spread of binades.
    Planes whose flatfield holds a zero, denormal, tiny or huge value, infinity or NaN never reach it
    (pre-pass), see the next test."""
    torch = _torch()
    dev = torch.device('cuda:0')
    for e0 in range(-100, 100, 40):
        assert native.selftest_flat_divide(e0, 40, False, dev) == 0, e0
    for e0 in (-100, -64, -20, -1, 0, 15, 16, 99):
        assert native.selftest_flat_divide(e0, 1, True, dev) == 0, e0
    with pytest.raises(native.NativeError, match='outside'):
        native.selftest_flat_divide(100, 1, False, dev)
    with pytest.raises(native.NativeError, match='outside'):
        native.selftest_flat_divide(-101, 1, False, dev)


def test_grouped_blend_division_equals_ieee_exhaustively():
    """Feather mode with plane groups divides the weighted sum by the weight sum with a reciprocal shared by the planes
    of the group: the IEEE sequence without its range handling.  Bit-equal to the compiler's division for every
    numerator (all 2^23 mantissas of the 97 binades 2^-44..2^52 that moderate gains can produce, both signs in a spread
    of them) and every weight sum 2..16384: 1.3e13 quotients."""
    torch = _torch()
    dev = torch.device('cuda:0')
    for e0 in range(-44, 53, 8):
        assert native.selftest_blend_divide(e0, min(8, 53 - e0), False, dev) == 0, e0
    for e0 in (-44, -1, 0, 16, 52):
        assert native.selftest_blend_divide(e0, 1, True, dev) == 0, e0
    with pytest.raises(native.NativeError, match='outside'):
        native.selftest_blend_divide(53, 1, False, dev)


def test_flatfield_fast_and_slow_paths_mix_in_one_vector():
    """Gains that leave the fast range (0, denormal, huge, inf, NaN, negative) sit next to ordinary
    ones inside the same 8-pixel vectors."""
    rng = np.random.default_rng(77)
    th, tw = 32, 64
    tiles = rng.integers(0, 65536, size=(1, 1, th, tw)).astype(np.uint16)
    flat = (0.5 + rng.random((th, tw))).astype(np.float32)
    odd = np.array([0.0, -0.0, 1e-40, 1e-38, 3e38, np.inf, -np.inf, np.nan, -1.5, 1e-31, 1e31, 2.0 ** -100, 2.0 ** 100,
                    2.0 ** -101, 1e-34, 5e-35], dtype=np.float32)
    for i, g in enumerate(odd):
        flat[(3 * i) % th, (11 * i + 5) % tw] = g
        flat[(5 * i + 1) % th, (7 * i) % tw] = g
    tiles[0, 0, 0, :8] = [0, 1, 65535, 2, 3, 4, 5, 6]
    rects = np.array([(0, 0, th, tw, 0, 0)])
    got, _ = run_fuse(rects, tiles, th, tw, flats_np=[flat])
    np.testing.assert_array_equal(got[0], O.fuse_plane_overwrite(list(tiles[0]), rects, th, tw, flat))


def test_fast_float64_divide_matches_ieee_on_random_gains():
    """float64 gains: the shortened sequence (the compiler's IEEE division without its range handling) gives
    the same double and the same clipped integer as the IEEE division, for 2^15 random gains per binade
    (plus all-zero, all-one and single-bit mantissas) x every uint16 numerator, over all 200 binades."""
    torch = _torch()
    dev = torch.device('cuda:0')
    for k, e0 in enumerate(range(-100, 100, 50)):
        assert native.selftest_flat_divide_f64(e0, 50, False, 1234 + k, dev) == 0, e0
    for e0 in (-100, -1, 0, 16, 99):
        assert native.selftest_flat_divide_f64(e0, 1, True, 99 + e0, dev) == 0, e0
    with pytest.raises(native.NativeError, match='outside'):
        native.selftest_flat_divide_f64(100, 1, False, 0, dev)


def test_rects_clipped_by_the_canvas_and_empty_rects():
    """Rectangles that stick out of the canvas are clipped like the reference's python slices
    (stitcher.py:590-594); rectangles that start outside it or have no area write nothing."""
    rng = np.random.default_rng(5)
    th, tw, ch, cw = 40, 56, 90, 100
    rects = np.array([(0, 0, th, tw, 70, 10),      # bottom rows clipped
                      (3, 5, 30, 50, 20, 80),      # right columns clipped
                      (0, 0, th, tw, 60, 60),      # both
                      (0, 0, th, tw, 90, 0),       # starts below the canvas
                      (0, 0, 0, 10, 5, 5),         # empty
                      (10, 10, 5, 0, 5, 5)])       # empty
    tiles = rng.integers(1, 65536, size=(1, len(rects), th, tw)).astype(np.uint16)
    got, plan = run_fuse(rects, tiles, ch, cw)
    want = O.fuse_plane_overwrite(list(tiles[0]), rects, ch, cw)
    np.testing.assert_array_equal(got[0], want)
    assert plan.covered_voxels == int((want != 0).sum())


@pytest.mark.parametrize('out_dtype', ['uint8', 'float32'])
def test_feather_uint8_tiles(out_dtype):
    rng = np.random.default_rng(8)
    th, tw, ch, cw = 33, 47, 120, 131
    rects = np.array([(0, 0, th, tw, int(y), int(x)) for y, x in zip(rng.integers(0, ch - 20, 12), rng.integers(0, cw - 20, 12))])
    tiles = rng.integers(0, 256, size=(2, 12, th, tw)).astype(np.uint8)
    flat = (0.7 + 0.6 * rng.random((th, tw))).astype(np.float64)
    got, _ = run_fuse(rects, tiles, ch, cw, mode=native.SQ_FUSE_FEATHER, flats_np=[flat, None], out_dtype=out_dtype, n_planes=2)
    for p, f in enumerate((flat, None)):
        np.testing.assert_array_equal(got[p], O.fuse_plane_feather(list(tiles[p]), rects, ch, cw, f, out_dtype=np.dtype(out_dtype).type))


@pytest.mark.parametrize('dtype', ['uint16', 'uint8'])
def test_fast_and_generic_divide_agree_on_real_planes(dtype):
    """Same tiles, same gains, once through the fast divide (every gain in range) and once through
    the generic IEEE sequence (one gain zeroed makes the pre-pass flag the plane): identical output
    wherever the zeroed gain is not involved, and both equal to the oracle."""
    rng = np.random.default_rng(123)
    th, tw, ch, cw, n = 72, 136, 300, 517, 14
    rects = random_rects(rng, n, th, tw, ch - 30, cw - 30)
    tiles = rng.integers(0, np.iinfo(dtype).max + 1, size=(1, n, th, tw)).astype(dtype)
    gains = np.exp(rng.normal(0.0, 1.5, size=(th, tw))).astype(np.float32)       # wide spread, all normal
    gains[5, 7], gains[9, 1], gains[40, 100] = 2.0 ** -99, 3.0e37, -0.75         # still inside the fast range
    fast, _ = run_fuse(rects, tiles, ch, cw, flats_np=[gains])
    np.testing.assert_array_equal(fast[0], O.fuse_plane_overwrite(list(tiles[0]), rects, ch, cw, gains))
    flagged = gains.copy()
    flagged[0, 0] = 0.0
    slow, _ = run_fuse(rects, tiles, ch, cw, flats_np=[flagged])
    want_slow = O.fuse_plane_overwrite(list(tiles[0]), rects, ch, cw, flagged)
    np.testing.assert_array_equal(slow[0], want_slow)
    same = want_slow == O.fuse_plane_overwrite(list(tiles[0]), rects, ch, cw, gains)
    np.testing.assert_array_equal(fast[0][same], slow[0][same])
    assert (~same).sum() <= n      # at most one voxel per tile differs (pixel (0, 0) where it is visible)


@pytest.mark.parametrize('queues', [False, True])
@pytest.mark.parametrize('seed', range(24))
def test_fuzz_geometry_dtype_flat_mode(seed, queues):
    """Random tile sizes (down to a few pixels: rows shorter than one 16-byte vector), canvas pitches,
    rectangle counts, dtypes, flatfield precisions, plane counts and both fusion modes; the kernels
    with their static walk (what launches this small take) and with the device work queues forced."""
    flags = native.SQ_FUSE_FORCE_QUEUES if queues else native.SQ_FUSE_FORCE_STATIC
    rng = np.random.default_rng(1000 + seed)
    dtype = ['uint16', 'uint8'][seed % 2]
    th, tw = int(rng.integers(1, 70)), int(rng.integers(1, 300))
    if seed % 5 == 0:
        tw = int(rng.integers(2049, 2300))                 # wider than one item: several column blocks
        th = int(rng.integers(1, 12))
    ch, cw = int(rng.integers(1, 200)), int(rng.integers(1, 700 if tw < 2049 else 2600))
    n = int(rng.integers(0, 12))
    planes = int(rng.integers(1, 4))
    rects = np.zeros((n, 6), dtype=np.int64)
    for i in range(n):
        sy, sx = int(rng.integers(0, th)), int(rng.integers(0, tw))
        rects[i] = (sy, sx, int(rng.integers(0, th - sy + 1)), int(rng.integers(0, tw - sx + 1)),
                    int(rng.integers(0, ch + 5)), int(rng.integers(0, cw + 5)))
    tiles = rng.integers(0, np.iinfo(dtype).max + 1, size=(planes, n, th, tw)).astype(dtype)
    flat_kind = seed % 3
    flats = None
    if flat_kind and n:
        fd = np.float32 if flat_kind == 1 else np.float64
        flats = []
        for p in range(planes):
            f = np.exp(rng.normal(0, 1, size=(th, tw))).astype(fd)
            if p % 2:
                f[rng.integers(0, th), rng.integers(0, tw)] = 0.0     # this plane takes the generic divide
            flats.append(f if p != 2 else None)
    feather = seed % 4 == 3 and n > 0
    if feather:
        out_dtype = [dtype, 'float32'][seed % 8 == 7]
        got, _ = run_fuse(rects, tiles, ch, cw, mode=native.SQ_FUSE_FEATHER, flats_np=flats, out_dtype=out_dtype, n_planes=planes,
                          flags=flags)
        for p in range(planes):
            want = O.fuse_plane_feather(list(tiles[p]), rects, ch, cw, None if flats is None else flats[p],
                                        out_dtype=np.dtype(out_dtype).type) if n else np.zeros((ch, cw), out_dtype)
            np.testing.assert_array_equal(got[p], want)
    else:
        got, plan = run_fuse(rects, tiles, ch, cw, flats_np=flats, n_planes=planes, flags=flags)
        for p in range(planes):
            want = O.fuse_plane_overwrite(list(tiles[p]), rects, ch, cw, None if flats is None else flats[p]) if n \
                else np.zeros((ch, cw), dtype)
            np.testing.assert_array_equal(got[p], want)


def test_planes_sharing_a_flatfield():
    """2 channels x 3 z planes: the planes of a channel share one gain image (pointer table with
    repeated entries)."""
    torch = _torch()
    rng = np.random.default_rng(31)
    th, tw, ch, cw, n = 40, 90, 150, 333, 11
    rects = random_rects(rng, n, th, tw, ch - 10, cw - 10)
    tiles = rng.integers(0, 65536, size=(6, n, th, tw)).astype(np.uint16)
    f0 = (0.5 + rng.random((th, tw))).astype(np.float32)
    f1 = (0.5 + rng.random((th, tw))).astype(np.float32)
    dev = torch.device('cuda:0')
    plan = native.FusePlan(rects, th, tw, ch, cw)
    d0, d1 = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
    canvas = torch.full((6, ch, cw), 9, dtype=torch.uint16, device=dev)
    native.fuse_planes(plan, torch.from_numpy(tiles).to(dev), canvas, [d0, d0, d0, d1, d1, d1])
    got = canvas.cpu().numpy()
    for p in range(6):
        np.testing.assert_array_equal(got[p], O.fuse_plane_overwrite(list(tiles[p]), rects, ch, cw, f0 if p < 3 else f1))


def test_padded_tile_and_canvas_pitches_through_the_c_abi():
    """tile_pitch > tile_w and canvas_pitch > canvas_w (the Python binding always passes dense buffers,
    the C-ABI does not require them): padding is neither read into the result nor written."""
    import ctypes as C
    torch = _torch()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(2)
    th, tw, tp, ch, cw, cp, n, planes = 33, 70, 77, 90, 151, 160, 6, 2
    rects = random_rects(rng, n, th, tw, ch - 5, cw - 5)
    tiles = rng.integers(0, 65536, size=(planes, n, th, tp)).astype(np.uint16)        # padded rows
    flat = (0.5 + rng.random((th, tw))).astype(np.float32)
    plan = native.FusePlan(rects, th, tw, ch, cw)
    d_tiles = torch.from_numpy(tiles).to(dev)
    d_canvas = torch.full((planes, ch, cp), 0xBEEF, dtype=torch.uint16, device=dev)
    d_flat = torch.from_numpy(flat).to(dev)
    fp = native.pointer_table([d_flat, d_flat], dev)
    table = plan.device_table(dev)
    L = native.lib()
    scratch = torch.empty(int(L.sq_fuse_scratch_bytes(planes)), dtype=torch.uint8, device=dev)
    a = native._FuseArgs()
    a.plan, a.table_dev, a.table_bytes = plan.handle, table.data_ptr(), table.numel()
    a.tile_base_dev, a.tile_plane_stride, a.tile_stride = d_tiles.data_ptr(), n * th * tp, th * tp
    a.n_tiles, a.tile_h, a.tile_w, a.tile_pitch, a.tile_dtype = n, th, tw, tp, native.SQ_U16
    a.flat_ptrs_dev, a.flat_dtype = fp.data_ptr(), native.SQ_F32
    a.canvas_dev, a.canvas_plane_stride, a.canvas_h, a.canvas_w, a.canvas_pitch = d_canvas.data_ptr(), ch * cp, ch, cw, cp
    a.canvas_dtype, a.n_planes, a.mode = native.SQ_U16, planes, native.SQ_FUSE_OVERWRITE
    a.scratch_dev, a.scratch_bytes = scratch.data_ptr(), scratch.numel()
    assert L.sq_fuse_planes(C.byref(a), native._stream_ptr()) == 0, L.sq_last_error()
    torch.cuda.synchronize()
    got = d_canvas.cpu().numpy()
    for p in range(planes):
        want = O.fuse_plane_overwrite(list(tiles[p][:, :, :tw]), rects, ch, cw, flat)
        np.testing.assert_array_equal(got[p][:, :cw], want)
        assert (got[p][:, cw:] == 0xBEEF).all()                 # the pitch padding is never written
    a.canvas_pitch = cw - 1
    assert L.sq_fuse_planes(C.byref(a), native._stream_ptr()) == -1 and b'pitch' in L.sq_last_error()


@pytest.mark.parametrize('planes', [1, 3])
def test_work_queues_on_a_registered_grid(planes):
    """The per-XCD queues on a lane-interleaved plan (registered 5x6 grid, cropped tiles, float32 gains, several
    planes): chunks that straddle planes, lanes of unequal length, the rest queue and stealing all occur."""
    from image_stitcher_amd import placement
    rng = np.random.default_rng(99)
    th, tw = 72, 200
    s = placement.Shifts((2, -31), (-19, -3))
    rects = placement.grid_rects(5, 6, tw, th, s)          # (rows, cols, width, height)
    wc, hc = placement.canvas_size(5, 6, tw, th, use_registration=True, shifts=s)
    tiles = rng.integers(0, 65536, size=(planes, 30, th, tw)).astype(np.uint16)
    flats = [(0.5 + rng.random((th, tw))).astype(np.float32) for _ in range(planes)]
    got, plan = run_fuse(rects, tiles, hc, wc, flats_np=flats, n_planes=planes, flags=native.SQ_FUSE_FORCE_QUEUES)
    assert plan.n_items > 64
    for p in range(planes):
        np.testing.assert_array_equal(got[p], O.fuse_plane_overwrite(list(tiles[p]), rects, hc, wc, flats[p]))


@pytest.mark.parametrize('mode', [native.SQ_FUSE_OVERWRITE, native.SQ_FUSE_FEATHER])
@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('queues', [False, True])
def test_plane_groups_on_line_aligned_canvases(seed, mode, queues):
    """Planes that share a gain image go through an item together (gains and reciprocals once per group) when the canvas
    planes start on 128-byte lines (native.empty_canvas).  Random registered-grid-like geometry, 1..12 planes dealt to 1..3
    gain images (one of them with a zero: that plane takes the generic divide in a group of its own), full groups of 5,
    partial groups, a plane without gains; both fusion modes, static walk and device queues: every plane equals the
    oracle.  The same planes on a DENSE stack (groups of one) give the same voxels."""
    torch = _torch()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(4200 + seed)
    th, tw = int(rng.integers(20, 70)), int(rng.integers(40, 260))
    rows, cols = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    oy, ox = int(rng.integers(2, th // 3)), int(rng.integers(2, tw // 3))
    n = rows * cols
    crop = mode == native.SQ_FUSE_OVERWRITE
    rects = np.zeros((n, 6), dtype=np.int64)
    for r in range(rows):
        for c in range(cols):
            top, left = (oy // 2 if r and crop else 0), (ox // 2 if c and crop else 0)
            bottom, right = (oy // 2 if r < rows - 1 and crop else 0), (ox // 2 if c < cols - 1 and crop else 0)
            rects[r * cols + c] = (top, left, th - top - bottom, tw - left - right,
                                   r * (th - oy) + top + c * 2, c * (tw - ox) + left + (rows - 1 - r) * 3)
    ch = int(rects[:, 4].max() + th + rng.integers(0, 9))
    cw = int(rects[:, 5].max() + tw + rng.integers(0, 9))
    planes = int(rng.integers(1, 13))
    tiles = rng.integers(0, 65536, size=(planes, n, th, tw)).astype(np.uint16)
    # overwrite mode, odd seeds: float64 gains (their plane groups share the float64 reciprocals)
    gdtype = np.float64 if (mode == native.SQ_FUSE_OVERWRITE and seed % 2) else np.float32
    gains = [np.exp(rng.normal(0, 0.4, size=(th, tw))).astype(gdtype) for _ in range(3)]
    gains[2][rng.integers(0, th), rng.integers(0, tw)] = 0.0            # this image needs the generic divide
    if mode == native.SQ_FUSE_FEATHER:
        gains[1][rng.integers(0, th), rng.integers(0, tw)] = 2.0 ** -30  # not moderate: no grouped blend for its planes
    which = [int(rng.integers(0, 3)) if rng.random() > 0.1 else -1 for _ in range(planes)]
    if planes >= 7:
        which[:6] = [0] * 6                                              # a full group of 5 and a leftover
    d_gains = [torch.from_numpy(g).to(dev) for g in gains]
    flats = [None if k < 0 else d_gains[k] for k in which]
    plan = native.FusePlan(rects, th, tw, ch, cw, mode)
    d_tiles = torch.from_numpy(tiles).to(dev)
    flags = native.SQ_FUSE_FORCE_QUEUES if queues else native.SQ_FUSE_FORCE_STATIC
    aligned = native.empty_canvas(planes, ch, cw, torch.uint16, dev)
    aligned.view(torch.int16).fill_(-1)
    dense = torch.full((planes, ch, cw), 7, dtype=torch.uint16, device=dev)
    if any(f is not None for f in flats):
        native.fuse_planes(plan, d_tiles, aligned, flats, flags=flags)
        native.fuse_planes(plan, d_tiles, dense, flats, flags=flags | native.SQ_FUSE_NO_PLANE_GROUPS)
    else:
        native.fuse_planes(plan, d_tiles, aligned, None, flags=flags)
        native.fuse_planes(plan, d_tiles, dense, None, flags=flags)
    torch.cuda.synchronize()
    # round 3: the planes of a key are DEALT to its groups round-robin (the default above); round 2's grouping -- five
    # consecutive planes of a key at a time -- must give the same voxels (SQ_FUSE_CONSECUTIVE_GROUPS, kept for A/B runs)
    consecutive = native.empty_canvas(planes, ch, cw, torch.uint16, dev)
    consecutive.view(torch.int16).fill_(-1)
    native.fuse_planes(plan, d_tiles, consecutive, flats if any(f is not None for f in flats) else None,
                       flags=flags | native.SQ_FUSE_CONSECUTIVE_GROUPS)
    torch.cuda.synchronize()
    for p in range(planes):
        g = None if which[p] < 0 else gains[which[p]]
        want = O.fuse_plane_overwrite(list(tiles[p]), rects, ch, cw, g) if mode == native.SQ_FUSE_OVERWRITE else \
            O.fuse_plane_feather(list(tiles[p]), rects, ch, cw, g, out_dtype=np.uint16)
        np.testing.assert_array_equal(aligned[p].cpu().numpy(), want, err_msg=f'plane {p} (gain image {which[p]}) of {planes}')
        np.testing.assert_array_equal(dense[p].cpu().numpy(), want, err_msg=f'dense plane {p}')
        np.testing.assert_array_equal(consecutive[p].cpu().numpy(), want, err_msg=f'plane {p}, groups of consecutive planes')


@pytest.mark.parametrize('gdtype', [np.float32, np.float64, None], ids=['f32', 'f64', 'nogain'])
@pytest.mark.parametrize('queues', [False, True], ids=['static', 'queues'])
@pytest.mark.parametrize('seed', range(4))
def test_uint8_planes_go_through_plane_groups(seed, queues, gdtype):
    """uint8 tiles (output dtype = input dtype, stitcher.py:228,362) through the plane groups (round 4): 16 pixels per lane and
    slot, the gains of a channel loaded once for up to five planes.  Registered-grid geometry wide enough for seam records (which
    a uint8 plane ignores as a whole: both neighbours write their part of a seam's line), 2..12 planes on 1..3 gain images (one
    holding a zero: the generic divide, groups of one) or none, canvas planes on 128-byte lines at an odd pitch: every voxel equals
    the oracle (truncation and the clip at 255 included), the per-plane kernel gives the same, the padding keeps its poison."""
    torch = _torch()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(8100 + seed)
    th, tw = int(rng.integers(24, 80)), int(rng.integers(150, 420))
    rows, cols = int(rng.integers(1, 4)), int(rng.integers(2, 4))
    oy, ox = int(rng.integers(2, th // 3)), int(rng.integers(4, tw // 5))
    n = rows * cols
    rects = np.zeros((n, 6), dtype=np.int64)
    for r in range(rows):
        for c in range(cols):
            top, left = (oy // 2 if r else 0), (ox // 2 if c else 0)
            bottom, right = (oy // 2 if r < rows - 1 else 0), (ox // 2 if c < cols - 1 else 0)
            rects[r * cols + c] = (top, left, th - top - bottom, tw - left - right, r * (th - oy) + top + c * 2, c * (tw - ox) + left + (rows - 1 - r) * 3)
    ch = int(rects[:, 4].max() + th + rng.integers(0, 9))
    cw = int(rects[:, 5].max() + tw + rng.integers(0, 9))
    planes = int(rng.integers(2, 13))
    tiles = rng.integers(0, 256, size=(planes, n, th, tw)).astype(np.uint8)
    flats = which = gains = None
    if gdtype is not None:
        gains = [np.exp(rng.normal(-0.3, 0.6, size=(th, tw))).astype(gdtype) for _ in range(3)]      # many quotients above 255: the clip
        gains[2][rng.integers(0, th), rng.integers(0, tw)] = 0.0
        which = [int(rng.integers(0, 3)) if rng.random() > 0.1 else -1 for _ in range(planes)]
        if planes >= 7:
            which[:6] = [0] * 6
        d_gains = [torch.from_numpy(g).to(dev) for g in gains]
        flats = [None if k < 0 else d_gains[k] for k in which]
        if not any(f is not None for f in flats):
            flats = None
    plan = native.FusePlan(rects, th, tw, ch, cw, native.SQ_FUSE_OVERWRITE)
    d_tiles = torch.from_numpy(tiles).to(dev)
    flags = native.SQ_FUSE_FORCE_QUEUES if queues else native.SQ_FUSE_FORCE_STATIC
    grouped = native.empty_canvas(planes, ch, cw, torch.uint8, dev)
    stride = grouped.stride(0)
    assert stride % 128 == 0
    backing = torch.full((planes * stride + 256,), 0x5A, dtype=torch.uint8, device=dev)
    grouped = backing[128:128 + planes * stride].as_strided((planes, ch, cw), (stride, cw, 1))
    single = native.empty_canvas(planes, ch, cw, torch.uint8, dev)
    native.fuse_planes(plan, d_tiles, grouped, flats, flags=flags)
    native.fuse_planes(plan, d_tiles, single, flats, flags=flags | native.SQ_FUSE_NO_PLANE_GROUPS)
    torch.cuda.synchronize()
    for p in range(planes):
        g = None if (which is None or which[p] < 0) else gains[which[p]]
        want = O.fuse_plane_overwrite(list(tiles[p]), rects, ch, cw, g)
        assert want.dtype == np.uint8
        np.testing.assert_array_equal(grouped[p].cpu().numpy(), want, err_msg=f'plane {p} of {planes} through the groups')
        np.testing.assert_array_equal(single[p].cpu().numpy(), want, err_msg=f'plane {p} through the per-plane kernel')
    host = backing.cpu().numpy()
    assert (host[:128] == 0x5A).all() and (host[128 + planes * stride:] == 0x5A).all()
    for p in range(planes):
        assert (host[128 + p * stride + ch * cw:128 + (p + 1) * stride] == 0x5A).all(), f'padding behind plane {p}'


@pytest.mark.gpu
@pytest.mark.parametrize('gdtype', [np.float32, np.float64, None], ids=['f32', 'f64', 'nogain'])
@pytest.mark.parametrize('queues', [False, True], ids=['static', 'queues'])
@pytest.mark.parametrize('seed', range(5))
def test_seam_lines_have_one_writer(seed, queues, gdtype):
    """Tiles at least a cache line wide either side of every vertical seam -- through the plane-group kernel, through the
    per-plane pipeline with the same gains (odd seeds: SQ_FUSE_NO_PLANE_GROUPS) and without gains: the item right of a seam writes
    the whole 128-byte line the seam falls in (pixels of BOTH tiles, or zero fill on the left), the item left of it stops
    at the line boundary (Seam in csrc/common.h).  Canvas pitches that put every row at a different phase, pitches that
    are multiples of a line, a base address off the line grid: every plane equals the oracle, and the canvas padding
    between rows and between planes keeps its poison."""
    torch = _torch()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5200 + seed)
    th, tw = int(rng.integers(12, 40)), int(rng.integers(150, 420))
    gr, gc = int(rng.integers(1, 4)), int(rng.integers(2, 5))
    rects = []
    x_margin = int(rng.integers(0, 2)) * int(rng.integers(64, 100))     # sometimes zero fill a line wide on the LEFT too
    for r in range(gr):
        for c in range(gc):
            left = int(rng.integers(0, 40)) if c else 0
            top = int(rng.integers(0, 6)) if r else 0
            rects.append((top, left, th - top, tw - left - int(rng.integers(0, 30)),
                          r * (th - 8) + int(rng.integers(0, 5)) + top, x_margin + c * (tw - 70) + int(rng.integers(0, 9)) + left))
    rects = np.array(rects)
    ch = int((rects[:, 4] + rects[:, 2]).max()) + int(rng.integers(0, 5))
    cw = int((rects[:, 5] + rects[:, 3]).max()) + int(rng.integers(0, 140))
    planes = int(rng.integers(2, 8))
    tiles = rng.integers(0, 65536, size=(planes, len(rects), th, tw)).astype(np.uint16)
    gain = None if gdtype is None else np.exp(rng.normal(0, 0.4, size=(th, tw))).astype(gdtype)
    d_gain = None if gain is None else torch.from_numpy(gain).to(dev)
    d_tiles = torch.from_numpy(tiles).to(dev)
    plan = native.FusePlan(rects, th, tw, ch, cw)
    flags = native.SQ_FUSE_FORCE_QUEUES if queues else native.SQ_FUSE_FORCE_STATIC
    if seed % 2:
        flags |= native.SQ_FUSE_NO_PLANE_GROUPS        # the per-plane pipeline honours the seam records too
    want = [O.fuse_plane_overwrite(list(tiles[p]), rects, ch, cw, gain) for p in range(planes)]
    for pitch, lead in ((cw, 0), (cw + 1, 0), (-(-cw // 64) * 64, 0), (cw + 37, 64 * 3), (cw + 2, 64 * 5 + 13)):
        stride = -(-(ch * pitch) // 64) * 64                 # planes a multiple of 128 bytes apart: groups form
        buf = torch.full((lead + planes * stride + 64,), 0x5A5A, dtype=torch.uint16, device=dev)
        canvas = buf[lead:].as_strided((planes, ch, cw), (stride, pitch, 1))
        native.fuse_planes(plan, d_tiles, canvas, None if d_gain is None else [d_gain] * planes, flags=flags)
        torch.cuda.synchronize()
        host = buf.cpu().numpy()
        for p in range(planes):
            got = host[lead + p * stride: lead + p * stride + ch * pitch].reshape(ch, pitch)
            np.testing.assert_array_equal(got[:, :cw], want[p], err_msg=f'plane {p}, pitch {pitch}')
            assert (got[:, cw:] == 0x5A5A).all(), f'row padding of plane {p} touched (pitch {pitch})'
            assert (host[lead + p * stride + ch * pitch: lead + (p + 1) * stride] == 0x5A5A).all()
        assert (host[:lead] == 0x5A5A).all() and (host[lead + planes * stride:] == 0x5A5A).all()


FLOAT_VOXEL_RTOL = 1e-5      # BASELINE.json north_star: "fused float voxels within 1e-5 relative"


def _assert_float_canvas_within_tolerance(got, want, what):
    """A float32 feather canvas from the grouped kernels: every voxel within FLOAT_VOXEL_RTOL of the definition, uncovered
    voxels (and whatever else the definition makes exactly 0) exactly 0; the largest deviation seen is in fact ~3e-7."""
    assert got.dtype == np.float32 and got.shape == want.shape
    zero = want == 0
    assert not got[zero].any(), f'{what}: voxels the definition leaves 0 are not 0'
    np.testing.assert_allclose(got[~zero], want[~zero], rtol=FLOAT_VOXEL_RTOL, atol=0, err_msg=what)
    with np.errstate(all='ignore'):
        worst = float(np.max(np.abs(got[~zero].astype(np.float64) - want[~zero]) / np.abs(want[~zero]), initial=0.0))
    assert worst < 2e-6, f'{what}: largest relative deviation {worst:.3g} -- within the tolerance but far above what the arithmetic explains'


@pytest.mark.parametrize('seed', range(4))
@pytest.mark.parametrize('queues', [False, True])
def test_feather_plane_groups_with_float64_gains(seed, queues):
    """Feather mode with float64 gain images through the plane groups (round 3): the blend takes a gain as float32 (the
    per-plane kernel casts every gain it loads), so the grouped blend loads the doubles and casts once per 8-pixel group.
    uint16 canvas on even seeds, float32 on odd ones; against the oracle and the per-plane kernel."""
    torch = _torch()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(6200 + seed)
    th, tw = int(rng.integers(20, 70)), int(rng.integers(40, 260))
    rows, cols = int(rng.integers(1, 4)), int(rng.integers(2, 4))
    oy, ox = int(rng.integers(2, th // 3)), int(rng.integers(2, tw // 3))
    n = rows * cols
    rects = np.zeros((n, 6), dtype=np.int64)
    for r in range(rows):
        for c in range(cols):
            rects[r * cols + c] = (0, 0, th, tw, r * (th - oy) + c * 2, c * (tw - ox) + (rows - 1 - r) * 3)
    ch = int(rects[:, 4].max() + th + rng.integers(0, 9))
    cw = int(rects[:, 5].max() + tw + rng.integers(0, 9))
    planes = int(rng.integers(2, 13))
    tiles = rng.integers(0, 65536, size=(planes, n, th, tw)).astype(np.uint16)
    gains = [np.exp(rng.normal(0, 0.4, size=(th, tw))).astype(np.float64) for _ in range(3)]
    gains[1][rng.integers(0, th), rng.integers(0, tw)] = 2.0 ** -30      # not moderate: groups of one for its planes
    which = [int(rng.integers(0, 3)) for _ in range(planes)]
    if planes >= 7:
        which[:6] = [0] * 6
    d_gains = [torch.from_numpy(g).to(dev) for g in gains]
    flats = [d_gains[k] for k in which]
    out_np = np.float32 if seed % 2 else np.uint16
    out_t = torch.float32 if seed % 2 else torch.uint16
    plan = native.FusePlan(rects, th, tw, ch, cw, native.SQ_FUSE_FEATHER)
    d_tiles = torch.from_numpy(tiles).to(dev)
    flags = native.SQ_FUSE_FORCE_QUEUES if queues else native.SQ_FUSE_FORCE_STATIC
    grouped = native.empty_canvas(planes, ch, cw, out_t, dev)
    single = native.empty_canvas(planes, ch, cw, out_t, dev)
    for c in (grouped, single):
        c.view(torch.int16 if out_t == torch.uint16 else torch.float32).fill_(-7)
    native.fuse_planes(plan, d_tiles, grouped, flats, flags=flags)
    native.fuse_planes(plan, d_tiles, single, flats, flags=flags | native.SQ_FUSE_NO_PLANE_GROUPS)
    torch.cuda.synchronize()
    for p in range(planes):
        want = O.fuse_plane_feather(list(tiles[p]), rects, ch, cw, gains[which[p]], out_dtype=out_np)
        for name, got in (('plane groups', grouped[p].cpu().numpy()), ('per-plane kernel', single[p].cpu().numpy())):
            if out_np is np.float32 and name == 'plane groups':
                _assert_float_canvas_within_tolerance(got, want, f'plane {p} of {planes}')
                continue
            ys, xs = np.nonzero(got != want)
            assert not len(ys), (f'{name}, plane {p} (gain image {which[p]}) of {planes}: {len(ys)} voxels differ, first '
                                 f'{[(int(y), int(x), float(got[y, x]), float(want[y, x])) for y, x in list(zip(ys, xs))[:10]]}')


@pytest.mark.parametrize('seed', range(4))
@pytest.mark.parametrize('queues', [False, True])
def test_feather_plane_groups_with_a_float32_canvas(seed, queues):
    """Feather mode into a float32 canvas with plane groups (round 3): the blended strips of the planes that share a
    gain image go through the grouped form (blend_item_zg<.., float, NREF>: strips, one-tile and empty items alike).  1..12
    planes on 1..3 gain images (one with a gain that is not moderate: groups of one for its planes) or none.  The per-plane
    kernel (SQ_FUSE_NO_PLANE_GROUPS) equals the oracle's float32 blend bit for bit; the grouped form (round 4) takes its
    quotients as n * (1 / g) and is held to the north star's tolerance for fused float voxels: 1e-5 relative, zeros exact."""
    torch = _torch()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5200 + seed)
    th, tw = int(rng.integers(20, 70)), int(rng.integers(40, 260))
    rows, cols = int(rng.integers(1, 4)), int(rng.integers(2, 4))
    oy, ox = int(rng.integers(2, th // 3)), int(rng.integers(2, tw // 3))
    n = rows * cols
    rects = np.zeros((n, 6), dtype=np.int64)
    for r in range(rows):
        for c in range(cols):
            rects[r * cols + c] = (0, 0, th, tw, r * (th - oy) + c * 2, c * (tw - ox) + (rows - 1 - r) * 3)
    ch = int(rects[:, 4].max() + th + rng.integers(0, 9))
    cw = int(rects[:, 5].max() + tw + rng.integers(0, 9))
    planes = int(rng.integers(2, 13))
    tiles = rng.integers(0, 65536, size=(planes, n, th, tw)).astype(np.uint16)
    gains = [np.exp(rng.normal(0, 0.4, size=(th, tw))).astype(np.float32) for _ in range(3)]
    gains[1][rng.integers(0, th), rng.integers(0, tw)] = 2.0 ** -30      # not moderate: no grouped blend for its planes
    use_gains = seed % 2 == 0
    which = [int(rng.integers(0, 3)) for _ in range(planes)] if use_gains else [-1] * planes
    if planes >= 7 and use_gains:
        which[:6] = [0] * 6                                               # a full group of 5 and a leftover
    d_gains = [torch.from_numpy(g).to(dev) for g in gains]
    flats = [d_gains[k] for k in which] if use_gains else None
    plan = native.FusePlan(rects, th, tw, ch, cw, native.SQ_FUSE_FEATHER)
    d_tiles = torch.from_numpy(tiles).to(dev)
    flags = native.SQ_FUSE_FORCE_QUEUES if queues else native.SQ_FUSE_FORCE_STATIC
    grouped = native.empty_canvas(planes, ch, cw, torch.float32, dev)
    grouped.fill_(-7.0)
    single = native.empty_canvas(planes, ch, cw, torch.float32, dev)
    single.fill_(-7.0)
    native.fuse_planes(plan, d_tiles, grouped, flats, flags=flags)
    native.fuse_planes(plan, d_tiles, single, flats, flags=flags | native.SQ_FUSE_NO_PLANE_GROUPS)
    torch.cuda.synchronize()
    for p in range(planes):
        want = O.fuse_plane_feather(list(tiles[p]), rects, ch, cw, gains[which[p]] if use_gains else None, out_dtype=np.float32)
        for name, got in (('plane groups', grouped[p].cpu().numpy()), ('per-plane kernel', single[p].cpu().numpy())):
            if name == 'plane groups':
                _assert_float_canvas_within_tolerance(got, want, f'plane {p} of {planes}')
                continue
            ys, xs = np.nonzero(got != want)
            assert not len(ys), (f'{name}, plane {p} (gain image {which[p]}) of {planes}, canvas {ch}x{cw}, tiles {th}x{tw}: {len(ys)} voxels differ, first '
                                 f'{[(int(y), int(x), float(got[y, x]), float(want[y, x])) for y, x in list(zip(ys, xs))[:10]]}')


@pytest.mark.parametrize('out_dtype', ['float32', 'uint16'])
def test_device_queues_lose_no_work_on_small_launches(out_dtype):
    """Regression test of the queue walk's LDS hazard (csrc/fuse.hip, for_each_queued_item / lds_written): a small
    feather plan (71 items x 12 planes, one or two work units per workgroup) through the per-plane kernel with the device
    queues, 300 launches, every one compared with the oracle on the device.  Before the fix 1-2 % of such launches left a
    wave's share of an item unwritten (profiles/r03_queue_stress_before_fix.log)."""
    torch = _torch()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5201)
    th, tw, rows, cols, oy, ox, planes = 29, 140, 3, 2, 5, 13, 12
    rects = np.zeros((rows * cols, 6), dtype=np.int64)
    for r in range(rows):
        for c in range(cols):
            rects[r * cols + c] = (0, 0, th, tw, r * (th - oy) + c * 2, c * (tw - ox) + (rows - 1 - r) * 3)
    ch, cw = int(rects[:, 4].max() + th + 2), int(rects[:, 5].max() + tw + 7)
    tiles = rng.integers(0, 65536, size=(planes, rows * cols, th, tw)).astype(np.uint16)
    plan = native.FusePlan(rects, th, tw, ch, cw, native.SQ_FUSE_FEATHER)
    d_tiles = torch.from_numpy(tiles).to(dev)
    tdt = torch.float32 if out_dtype == 'float32' else torch.uint16
    want = np.stack([O.fuse_plane_feather(list(tiles[p]), rects, ch, cw, None, out_dtype=np.dtype(out_dtype)) for p in range(planes)])
    d_want = torch.from_numpy(want.astype(np.float32)).to(dev)
    wrong = []
    for it in range(300):
        canvas = native.empty_canvas(planes, ch, cw, tdt, dev)
        canvas.view(torch.int16 if tdt == torch.uint16 else torch.float32).fill_(-7)
        native.fuse_planes(plan, d_tiles, canvas, None, flags=native.SQ_FUSE_FORCE_QUEUES | native.SQ_FUSE_NO_PLANE_GROUPS)
        n = int((canvas.to(torch.float32) != d_want).sum())
        if n:
            wrong.append((it, n))
    assert not wrong, f'launches with wrong voxels (launch, voxels): {wrong[:10]}'

"""Pyramid kernel (sq_downsample2) and the streamed OME-Zarr output on the GPU,
against the oracle's restatement of ome_zarr's Scaler.nearest (pinned in test_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_case, spec_of, flatfields_for
from image_stitcher_amd import native, omezarr, synth
from image_stitcher_amd.stitcher import Stitcher
from image_stitcher_amd.stitcher_parameters import StitchingParameters
from oracle import stitch_oracle as O

pytestmark = pytest.mark.gpu


def test_downsample2_matches_third_party_vectors():
    v = np.load(os.path.join(GOLDEN, 'pyramid_vectors.npz'))
    for key in v.files:
        if not key.startswith('in_'):
            continue
        src = torch.from_numpy(v[key][None].copy()).cuda()
        got = native.downsample2(src).cpu().numpy()[0]
        np.testing.assert_array_equal(got, v['zoom_' + key[3:]])


@pytest.mark.parametrize('dtype', ['uint16', 'uint8'])
@pytest.mark.parametrize('shape', [(3, 1001, 777), (2, 64, 4096), (1, 2, 2), (1, 3, 1030), (2, 517, 33), (1, 1, 40), (1, 40, 1),
                                   (1, 2300, 5000)])
def test_downsample2_shapes(dtype, shape):
    rng = np.random.default_rng(sum(shape))
    a = rng.integers(0, np.iinfo(dtype).max + 1, shape).astype(dtype)
    got = native.downsample2(torch.from_numpy(a).cuda())
    assert tuple(got.shape) == (shape[0], shape[1] // 2, shape[2] // 2)
    np.testing.assert_array_equal(got.cpu().numpy(), O.pyramid_nearest(a, 2)[1] if min(shape[1:]) >= 2
                                  else np.zeros((shape[0], shape[1] // 2, shape[2] // 2), dtype))


def test_downsample2_pitched_views_and_guard_bytes():
    """Source and destination are windows of larger buffers at odd element offsets: every phase of the
    16-byte store alignment, and nothing outside the destination window is touched."""
    rng = np.random.default_rng(3)
    big = torch.from_numpy(rng.integers(0, 65536, (2, 300, 700)).astype(np.uint16)).cuda()
    for off in range(0, 9):
        src = big[:, 3:3 + 201, off:off + 403]
        out_big = torch.full((2, 120, 260), 0xABCD, dtype=torch.uint16).cuda()
        dst = out_big[:, 5:5 + 100, off + 1:off + 1 + 201]
        native.downsample2(src, out=dst)
        want = O.pyramid_nearest(src.cpu().numpy(), 2)[1]
        got = out_big.cpu().numpy()
        np.testing.assert_array_equal(got[:, 5:105, off + 1:off + 202], want)
        got[:, 5:105, off + 1:off + 202] = 0xABCD
        assert (got == 0xABCD).all()
    with pytest.raises(ValueError):
        native.downsample2(big, out=torch.empty((2, 150, 351), dtype=torch.uint16).cuda())
    with pytest.raises(ValueError):
        native.downsample2(big.cpu())


def test_downsample2_chain_equals_strided_gather_of_level0():
    """Level k samples level 0 at 2^k * o + 2^k - 1: a size-independent property, at a full-size plane."""
    g = torch.Generator(device='cuda').manual_seed(1)
    a = torch.randint(0, 65536, (1, 9107, 7277), device='cuda', generator=g, dtype=torch.int32).to(torch.uint16)
    lv = a
    for k in range(1, 6):
        lv = native.downsample2(lv)
        step = 2 ** k
        want = a.view(torch.int16)[:, step - 1::step, step - 1::step][:, :lv.shape[1], :lv.shape[2]]
        assert tuple(lv.shape) == (1, 9107 >> k, 7277 >> k)
        assert torch.equal(lv.view(torch.int16), want)


def test_plane_stream_writer_multi_batch(tmp_path):
    rng = np.random.default_rng(8)
    img = rng.integers(0, 65536, (1, 2, 3, 333, 1201)).astype(np.uint16)
    img[0, 0, 1] = 0
    levels = O.pyramid_nearest(img, 4)
    for compression in ('zlib', 'none'):
        path = str(tmp_path / f's_{compression}.ome.zarr')
        shapes = omezarr.create_store(path, img.shape, img.dtype, pixel_size_um=0.5, num_levels=4, chunks=(1, 1, 1, 128, 256),
                                      compression=compression)
        coords = [(0, c, z) for c in range(2) for z in range(3)]
        planes = torch.from_numpy(img.reshape(6, 333, 1201)).cuda()
        with omezarr.PlaneStreamWriter(path, shapes, img.dtype, chunks=(1, 1, 1, 128, 256), batch=2, compression=compression,
                                       device=planes.device) as w:
            for b0 in (0, 2, 4):          # three batches through two slots
                w.acquire(2).copy_(planes[b0:b0 + 2])
                w.submit(coords[b0:b0 + 2])
            with pytest.raises(ValueError):
                w.acquire(3)
        assert w.bytes_written > 0
        for lv in range(4):
            np.testing.assert_array_equal(omezarr.read_array(os.path.join(path, str(lv))), levels[lv])
    # a device tensor or a numpy array through the one-call writer
    for k, image in enumerate((torch.from_numpy(img).cuda(), img)):
        path = omezarr.write_ome_zarr(str(tmp_path / f'w{k}.ome.zarr'), image, pixel_size_um=0.5, num_levels=3)
        for lv in range(3):
            np.testing.assert_array_equal(omezarr.read_array(os.path.join(path, str(lv))), levels[lv])


def test_streamed_region_equals_in_memory_region(tmp_path):
    """run()'s streaming path writes the store save_region_ome_zarr(stitch_region()) writes, and level 0 is
    the reference's canvas; batches of one plane force every slot to be reused."""
    info, arrays = load_case('reg_ff32')
    root = str(tmp_path / 'acq')
    synth.write_acquisition(spec_of(info), root)
    p = info['params']

    def prepared(out):
        st = Stitcher(StitchingParameters(input_folder=root, use_registration=True, apply_flatfield=True,
                                          registration_channel=p['registration_channel'],
                                          registration_z_level=p['registration_z_level']), normalization=None)
        st.output_folder = str(tmp_path / out)
        st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
        st.flatfields = flatfields_for(info, st.num_c)
        st.calculate_shifts(0, 'R0')
        dims = st.calculate_output_dimensions

        def forced(t, r):       # small canvases have one level; ask for three
            wh = dims(t, r)
            st.num_pyramid_levels = 3
            return wh
        st.calculate_output_dimensions = forced
        return st

    a = prepared('mem')
    path_a = a.save_region_ome_zarr(0, 'R0', a.stitch_region(0, 'R0', device_output=True))
    b = prepared('stream')
    b.batch_bytes_limit = 1
    seen = []
    path_b = b.stream_region_to_zarr(0, 'R0', progress_callback=lambda i, n: seen.append((i, n)))
    assert len(seen) == len(b.get_region_data(0, 'R0')) and b.last_bytes_written > 0
    want = O.pyramid_nearest(arrays['t0_R0_canvas'], 3)
    for lv in range(3):
        got = omezarr.read_array(os.path.join(path_b, str(lv)))
        np.testing.assert_array_equal(got, omezarr.read_array(os.path.join(path_a, str(lv))))
        np.testing.assert_array_equal(got, want[lv])
    # a subset of planes (what one rank of a plane-shared region writes)
    c = prepared('part')
    c.create_region_store(0, 'R0')
    c.stream_region_to_zarr(0, 'R0', only_planes=[1], create=False)
    got = omezarr.read_array(os.path.join(c._zarr_path(0, 'R0'), '1'))
    np.testing.assert_array_equal(got[0, 1, 0], want[1][0, 1, 0])
    assert not got[0, 0].any()

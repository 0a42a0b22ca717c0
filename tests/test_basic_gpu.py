"""Flatfield estimate on the device (csrc/basic.hip) against its CPU definition (oracle/basic_oracle.py) and a planted
gain.  Parity with the reference's basicpy call is UNPINNED (the package is absent offline); what is tested is that
the device runs the algorithm the oracle defines, and that the algorithm does its job."""
import numpy as np
import pytest

from image_stitcher_amd import native, synth
from oracle import basic_oracle as B
from test_basic_oracle_cpu import planted_stack

pytestmark = pytest.mark.gpu


def _dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('n,h,w,dtype', [(40, 256, 320, 'uint16'), (12, 96, 200, 'uint8'), (33, 2048, 2048, 'uint16')])
def test_device_fit_matches_the_definition_and_the_planted_gain(n, h, w, dtype):
    import torch
    # a sparse foreground is BaSiC's premise: a handful of blobs on the small tiles (30 would cover a fifth of them)
    stack, gain = planted_stack(n, h, w, seed=n + h, objects=6 if h < 128 else 30 * max(1, (h * w) // (256 * 320)) if h > 256 else 30)
    if dtype == 'uint8':
        stack = (stack >> 6).astype(np.uint8)
    flat_dev, info = native.basic_fit(torch.from_numpy(stack).to(_dev()))
    flat = flat_dev.cpu().numpy()
    want, winfo = B.basic_fit(stack)
    assert flat.shape == (h, w) and flat.dtype == np.float32 and info['working_size'] == 128
    # same algorithm: float32 with different summation orders, a few hundred iterations -> 2e-3
    assert np.abs(flat / want - 1.0).max() < 2e-3, (np.abs(flat / want - 1.0).max(), info, winfo['ladmap_iterations'])
    assert abs(info['ladmap_iterations'] - sum(winfo['ladmap_iterations'])) <= 2 * len(winfo['ladmap_iterations'])
    err = np.abs(flat / flat.mean() / gain - 1.0)
    if h == 256:      # the planted gain, where the blobs survive the resampling to 128 x 128 as sparse foreground
        assert err.mean() < 2e-3 and np.quantile(err, 0.999) < 1e-2, (err.mean(), err.max())
    else:
        assert err.mean() < 1e-2, err.mean()


def test_resampling_kernels_equal_the_definition():
    """One 'image' of every pixel value pattern: the device resize (down to 128 x 128 and back up) follows
    oracle.resize to float32 rounding -- checked through a fit of a single-image stack whose gain is the image."""
    import torch
    rng = np.random.default_rng(1)
    img = (2000 + 1000 * synth.synthetic_flatfield(300, 517, np.float32)).astype(np.uint16)
    stack = np.stack([img, img, img])
    flat, _ = native.basic_fit(torch.from_numpy(stack).to(_dev()))
    want, _ = B.basic_fit(stack)
    assert np.abs(flat.cpu().numpy() / want - 1.0).max() < 1e-3


def test_bad_arguments():
    import torch
    with pytest.raises(native.NativeError, match='images'):
        native.basic_fit(torch.zeros((65, 32, 32), dtype=torch.uint16, device=_dev()))
    with pytest.raises(native.NativeError, match='all zero'):
        native.basic_fit(torch.zeros((4, 32, 32), dtype=torch.uint16, device=_dev()))
    with pytest.raises(ValueError):
        native.basic_fit(torch.zeros((4, 32, 32), dtype=torch.uint16))

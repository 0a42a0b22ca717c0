"""Shared helpers for the test-suite (fixtures loading, acquisition rebuilds)."""
import glob
import hashlib
import json
import os

import numpy as np

import image_stitcher_amd.synth as synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

REGION_CASES = sorted(os.path.splitext(os.path.basename(p))[0]
                      for p in glob.glob(os.path.join(GOLDEN, '*.json'))
                      if not os.path.basename(p).startswith(('pcc_', 'blosc_')))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_case(name):
    with open(os.path.join(GOLDEN, name + '.json')) as fh:
        info = json.load(fh)
    arrays = np.load(os.path.join(GOLDEN, name + '.npz'))
    return info, arrays


def spec_of(info):
    d = dict(info['spec'])
    for k in ('channels', 'regions', 'rgb_channels', 'blank_fovs'):
        d[k] = tuple(d.get(k, ()))
    d['missing'] = tuple(tuple(m) for m in d.get('missing', ()))
    d['region_dims'] = tuple(tuple(m) for m in d.get('region_dims', ()))
    return synth.GridSpec(**d)


def flatfields_for(info, n_channels):
    """The synthetic flatfields make_golden.py assigned (same formula)."""
    p = info['params']
    if not p['apply_flatfield']:
        return None
    dt = np.dtype(p['flat_dtype'])
    sp = info['spec']
    out = {}
    for ci in range(n_channels):
        ff = synth.synthetic_flatfield(sp['tile_h'], sp['tile_w'], dt)
        out[ci] = (ff * dt.type(1.0 + 0.03125 * ci)).astype(dt)
    return out


# Cases whose crops do not contain the true overlap (pixel_binning 1 halves the crop width): the shifts are
# whatever the correlation makes of unrelated crops, so the two normalisation modes need not agree there.
PHASE_MAY_DIFFER = {'reg_binning1', 'reg_binning1_odd'}

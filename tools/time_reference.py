#!/opt/conda/bin/python3.9
"""Time the GENUINE reference (unmodified /root/reference/stitcher.py) on this container's host cores,
next to this repo's numpy oracle on the same inputs.  Authoring container only (the reference never
travels): `/opt/conda/bin/python3.9 tools/time_reference.py`.  Same stand-ins for the six I/O-only
packages as tests/golden/make_golden.py; nothing of the reference is copied.  Results are quoted in
DESIGN.md section 6 (they are not part of bench.py: the GPU box has no reference)."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
sys.path.insert(0, REPO)
import make_golden as G          # noqa: E402  installs the stand-ins and imports the reference
from image_stitcher_amd import synth   # noqa: E402
from oracle import stitch_oracle as O  # noqa: E402

ref = G.ref_stitcher


def run(name, spec, use_registration, apply_flatfield):
    tmp = tempfile.mkdtemp(prefix='tref_')
    root = os.path.join(tmp, 'acq')
    try:
        t0 = time.time()
        synth.write_acquisition(spec, root)
        t_write = time.time() - t0
        params = G.StitchingParameters(input_folder=root, use_registration=use_registration, apply_flatfield=apply_flatfield)
        st = ref.Stitcher(params)
        t0 = time.time()
        st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
        t_meta = time.time() - t0
        if apply_flatfield:
            for ci in range(st.num_c):
                st.flatfields[ci] = synth.synthetic_flatfield(spec.tile_h, spec.tile_w, np.float32)
        t_reg = 0.0
        if use_registration:
            t0 = time.time()
            st.calculate_shifts(st.timepoints[0], st.regions[0])
            t_reg = time.time() - t0
        t0 = time.time()
        canvas = st.stitch_region(int(st.timepoints[0]), st.regions[0])
        canvas = np.asarray(canvas.compute() if hasattr(canvas, 'compute') else canvas)
        t_fuse = time.time() - t0
        vox = canvas.size
        # the oracle on the same files
        t0 = time.time()
        from image_stitcher_amd.tiffio import read_image
        acq = O.parse_acquisition(root, read_image)
        shifts = None
        if use_registration:
            shifts = O.calculate_shifts(acq, acq.timepoints[0], acq.regions[0], read_image, normalization=None)
        t_oreg = time.time() - t0
        t0 = time.time()
        flats = {ci: synth.synthetic_flatfield(spec.tile_h, spec.tile_w, np.float32) for ci in range(len(acq.channel_names))} \
            if apply_flatfield else None
        got = O.stitch_region(acq, acq.timepoints[0], acq.regions[0], read_image, use_registration=use_registration,
                              shifts=shifts, flatfields=flats, apply_flat=apply_flatfield)
        t_ofuse = time.time() - t0
        same = bool(np.array_equal(np.asarray(got), canvas))
        print(f'{name}: tiles written {t_write:.1f}s | reference: metadata {t_meta:.2f}s, calculate_shifts {t_reg:.2f}s, '
              f'stitch_region.compute() {t_fuse:.2f}s = {vox / t_fuse / 1e6:.1f} Mvoxel/s ({vox / (t_fuse + t_reg) / 1e6:.1f} incl. registration) '
              f'| oracle: shifts {t_oreg:.2f}s, stitch_region {t_ofuse:.2f}s = {vox / t_ofuse / 1e6:.1f} Mvoxel/s, identical canvas: {same} '
              f'| host cores: {os.cpu_count()}', flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


S = synth.GridSpec
which = sys.argv[1:] or ['cfg1', 'cfg2', 'cfg3plane']
if 'cfg1' in which:
    run('config 1 (2x2 x 512^2, coordinate-only)', S(rows=2, cols=2, tile_h=512, tile_w=512, ov_y=77, ov_x=77, seed=1000), False, False)
if 'cfg2' in which:
    run('config 2 (8x8 x 2048^2, -r)', S(rows=8, cols=8, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=2000), True, False)
if 'cfg3plane' in which:
    run('one (c,z) plane of config 3 (16x16 x 2048^2, -r -ff)', S(rows=16, cols=16, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=3000), True, True)

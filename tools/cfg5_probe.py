"""BASELINE config 5 in small: W wells of 5x5 tiles of 2048^2, 3 channels, T timepoints, per-well registration, streamed to OME-Zarr
through the CLI's own path (Stitcher.run), acquisition and stores on /dev/shm: seconds per (well, timepoint) unit and where they go.
    python tools/cfg5_probe.py [wells=8] [T=2]"""
import contextlib, io, os, shutil, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import synth
from image_stitcher_amd.stitcher import Stitcher
from image_stitcher_amd.stitcher_parameters import StitchingParameters

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
wells = [f'{"ABCDEFGH"[i // 12]}{i % 12 + 1}' for i in range(W)]
spec = synth.GridSpec(rows=5, cols=5, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=5000, channels=synth.DEFAULT_CHANNELS[:3], nz=1, nt=T,
                      regions=tuple(wells))
tmp = tempfile.mkdtemp(prefix='cfg5_', dir='/dev/shm')
try:
    root = os.path.join(tmp, 'acq')
    t0 = time.time()
    synth.write_acquisition_device(spec, root, dev)
    print(f'{W} wells x {T} timepoints x 75 files written in {time.time() - t0:.1f} s', flush=True)
    for rep in range(2):
        st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), per_region_registration=True)
        buf = io.StringIO()
        t0 = time.time()
        with contextlib.redirect_stdout(buf):
            st.run()
        dt = time.time() - t0
        log = buf.getvalue()
        per = [float(l.rsplit(':', 1)[1]) for l in log.splitlines() if l.startswith('Completed region')]
        stitch = [float(l.rsplit(':', 1)[1]) for l in log.splitlines() if l.startswith('Time to stitch region')]
        w, h = st.calculate_output_dimensions(0, st.regions[0])
        vox = 3 * w * h
        print(f'run {rep}: {W * T} units in {dt:.2f} s = {dt / (W * T) * 1e3:.0f} ms per unit ({vox * W * T / dt / 1e9:.2f} Gvoxel/s); per unit: completed {np.mean(per) * 1e3:.0f} ms '
              f'(min {min(per) * 1e3:.0f}, max {max(per) * 1e3:.0f}), of which stitch_planes {np.mean(stitch) * 1e3:.0f} ms; metadata + flatfields + the rest {1e3 * (dt - sum(per)):.0f} ms', flush=True)
        shutil.rmtree(st.output_folder, ignore_errors=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)

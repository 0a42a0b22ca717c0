"""End-to-end CLI run from files on the GPU box: write a synthetic acquisition, run the drop-in CLI,
report the phases (not the bench contract; disk + zlib dominate)."""
import os, sys, time, tempfile, shutil, io, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import synth, stitcher_cli

def run(name, spec, argv):
    tmp = tempfile.mkdtemp(prefix='e2e_', dir=os.environ.get('TMPDIR', '/tmp'))
    try:
        root = os.path.join(tmp, 'acq')
        t0 = time.time(); synth.write_acquisition(spec, root); t_write = time.time() - t0
        buf = io.StringIO()
        t0 = time.time()
        with contextlib.redirect_stdout(buf):
            stitcher_cli.main(['-i', root] + argv)
        total = time.time() - t0
        log = buf.getvalue()
        stitch = [float(l.split(':')[-1]) for l in log.splitlines() if l.startswith('Time to stitch region')]
        out = [d for d in os.listdir(tmp) if d.startswith('acq_stitched_')][0]
        nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(tmp, out)) for f in fs)
        vox = None
        for l in log.splitlines():
            if 'output array dimensions' in l:
                shape = eval(l.split(':')[-1]); vox = int(np.prod(shape))
        print(f'{name}: generate+write tiles {t_write:.1f}s | CLI total {total:.1f}s of which stitch_region (read files, H2D, '
              f'register+fuse, D2H) {sum(stitch):.2f}s -> {vox/sum(stitch)/1e6:.0f} Mvoxel/s; store {nbytes/1e6:.0f} MB', flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

G = synth.GridSpec
run('config 1 (2x2 x 512^2, coordinate-only)', G(rows=2, cols=2, tile_h=512, tile_w=512, ov_y=77, ov_x=77, seed=1000), [])
run('config 2 (8x8 x 2048^2, -r)', G(rows=8, cols=8, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=2000), ['-r'])


def stream_vs_sequential(spec, batch_bytes):
    """Files -> OME-Zarr store for one region, two ways: fuse everything, copy the region to the host, then
    write (what stitch_region + save_region_ome_zarr do), against the streamed path run() uses."""
    from image_stitcher_amd.stitcher import Stitcher
    from image_stitcher_amd.stitcher_parameters import StitchingParameters
    tmp = tempfile.mkdtemp(prefix='e2e_', dir=os.environ.get('TMPDIR', '/tmp'))
    try:
        root = os.path.join(tmp, 'acq')
        import torch
        t0 = time.time(); synth.write_acquisition_device(spec, root, torch.device('cuda:0')); t_write = time.time() - t0
        print(f'stream probe: {spec.rows}x{spec.cols} x {spec.tile_h}^2, {len(spec.channels)} ch x {spec.nz} z; tiles written in {t_write:.1f}s', flush=True)
        for compression in ('blosc', 'zlib', 'none'):
            for mode in (('streamed',) if compression == 'blosc' else ('sequential', 'streamed')):
                st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), zarr_compression=compression)
                st.output_folder = os.path.join(tmp, f'out_{compression}_{mode}')
                st.batch_bytes_limit = batch_bytes
                with contextlib.redirect_stdout(io.StringIO()):
                    st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
                    st.calculate_shifts(0, st.regions[0])
                    t0 = time.time()
                    if mode == 'sequential':
                        region = st.stitch_region(0, st.regions[0])
                        t_fuse = time.time() - t0
                        path = st.save_region_ome_zarr(0, st.regions[0], region)
                        vox = region.size
                    else:
                        path = st.stream_region_to_zarr(0, st.regions[0])
                        t_fuse = float('nan')
                        w, h = st.calculate_output_dimensions(0, st.regions[0])
                        vox = st.num_c * st.num_z * w * h
                    total = time.time() - t0
                nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(path) for f in fs)
                print(f'  {compression:5s} {mode:10s}: {total:6.2f}s files->store ({vox / total / 1e6:7.0f} Mvoxel/s), '
                      f'stitch_region alone {t_fuse:.2f}s, levels {st.num_pyramid_levels}, store {nbytes / 1e6:.0f} MB', flush=True)
                shutil.rmtree(st.output_folder, ignore_errors=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


stream_vs_sequential(G(rows=16, cols=16, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=2500, nz=4), batch_bytes=256 * 2048 * 2048 * 2)

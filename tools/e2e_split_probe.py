"""End to end from files, taken apart: where the seconds of a files -> six-level OME-Zarr run go, and whether the disk or the
pipeline is the wall.  Four config-3 planes (16x16 x 2048^2, 1 ch x 4 z: 1024 files, 8.6 GB in), once with the acquisition and the
store on the directory given (the box's disk) and once on /dev/shm:
    read      files -> page-locked staging by the decode threads alone (no device)                       GB/s of pixels
    h2d       page-locked staging -> device, the copies alone (the PCIe rate the ingest could reach)      GB/s
    ingest    files -> device canvas: Stitcher.stitch_region(device_output=True) (read + H2D + fusion)    GB/s of pixels, s
    encode    pyramid levels + Blosc-1 chunk frames of the fused planes, on the device                    GB/s of pixels in
    write     the frames -> files from memory: Python's file calls, then native threads (sq_write_files)      GB/s of frame bytes
    run       files -> store, streamed (Stitcher.stream_region_to_zarr)                                   s, Gvoxel/s
    python tools/e2e_split_probe.py [dir=/tmp] [z=4]"""
import contextlib, io, os, shutil, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, omezarr, synth, tiffio
from image_stitcher_amd.stitcher import Stitcher
from image_stitcher_amd.stitcher_parameters import StitchingParameters
from concurrent.futures import ThreadPoolExecutor

base = sys.argv[1] if len(sys.argv) > 1 else '/tmp'
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
spec = synth.GridSpec(rows=16, cols=16, tile_h=2048, tile_w=2048, ov_y=244, ov_x=244, seed=2500, nz=nz)
T = 2048


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def probe(where):
    tmp = tempfile.mkdtemp(prefix='e2e_', dir=where)
    try:
        root = os.path.join(tmp, 'acq')
        t0 = time.time()
        synth.write_acquisition_device(spec, root, dev)
        files = sorted(os.path.join(root, '0', f) for f in os.listdir(os.path.join(root, '0')) if f.endswith('.tiff'))
        in_bytes = len(files) * T * T * 2
        print(f'[{where}] {len(files)} files, {in_bytes / 1e9:.1f} GB of pixels, written in {time.time() - t0:.1f} s', flush=True)
        # ---- read: files -> pinned staging, decode threads alone
        staging = torch.empty((len(files), T, T), dtype=torch.uint16, pin_memory=True)
        host = staging.numpy()
        for workers in (16, 32):
            with ThreadPoolExecutor(max_workers=workers) as pool:
                t0 = time.time()
                ok = list(pool.map(lambda k: tiffio.read_image_into(files[k], host[k]), range(len(files))))
                dt = time.time() - t0
            assert all(ok)
            print(f'[{where}] read    {workers:2d} threads: {dt:6.2f} s  {in_bytes / dt / 1e9:6.2f} GB/s  (files in the page cache after the write)', flush=True)
        # ---- h2d: the copies alone
        on_dev = torch.empty((len(files), T, T), dtype=torch.uint16, device=dev)
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            on_dev.copy_(staging, non_blocking=True)
            torch.cuda.synchronize(); dt = time.time() - t0
        print(f'[{where}] h2d     one copy of the staging: {dt:6.2f} s  {in_bytes / dt / 1e9:6.2f} GB/s', flush=True)
        del on_dev, staging, host
        # ---- ingest: files -> device canvas through the product's pipeline
        st = Stitcher(StitchingParameters(input_folder=root, use_registration=True), zarr_compression='blosc')
        st.output_folder = os.path.join(tmp, 'out')
        with quiet():
            st.get_timepoints(); st.extract_acquisition_parameters(); st.get_pixel_size(); st.parse_acquisition_metadata()
            st.calculate_shifts(0, st.regions[0])
        for rep in range(2):
            with quiet():
                torch.cuda.synchronize(); t0 = time.time()
                region = st.stitch_region(0, st.regions[0], device_output=True)
                torch.cuda.synchronize(); dt = time.time() - t0
            vox = region.numel()
            print(f'[{where}] ingest  files -> device canvas (run {rep}): {dt:6.2f} s  {in_bytes / dt / 1e9:6.2f} GB/s of pixels, {vox / dt / 1e9:5.2f} Gvoxel/s'
                  + (f'  [canvas arena: {st.canvas_arena_info["class_slices"]}, {st.canvas_arena_info["create_ms"]:.0f} ms]' if rep == 0 and st.canvas_arena_info else ''), flush=True)
            if rep == 0:
                del region
        # ---- encode: pyramid + chunk frames on the device
        planes = region[0].flatten(0, 1)          # [C * Z, Hc, Wc] view
        levels = [planes]
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(st.num_pyramid_levels - 1):
            levels.append(native.downsample2(levels[-1]))
        encs = [native.blosc_encode_planes(lv, 512, 512) for lv in levels]
        torch.cuda.synchronize(); dt = time.time() - t0
        px = sum(lv.numel() for lv in levels) * 2
        frames = []
        for e in encs:
            off = e.offsets.cpu().numpy()
            frames.append((e.out[:int(off[-1])].cpu().numpy(), off))
        fbytes = sum(int(o[-1]) for _, o in frames)
        print(f'[{where}] encode  {len(levels)} levels, {px / 1e9:.1f} GB of pixels -> {fbytes / 1e9:.2f} GB of frames: {dt:6.2f} s  {px / dt / 1e9:6.1f} GB/s of pixels in', flush=True)
        # ---- write: the frames to files, as the writer threads do (one file per non-empty chunk)
        outdir = os.path.join(tmp, 'frames')
        os.makedirs(outdir)
        jobs = [(li, i, buf[o[i]:o[i + 1]]) for li, (buf, o) in enumerate(frames) for i in range(len(o) - 1) if o[i + 1] > o[i]]

        def put(job):
            li, i, b = job
            with open(os.path.join(outdir, f'{li}.{i}'), 'wb') as fh:
                fh.write(b)
        with ThreadPoolExecutor(max_workers=16) as pool:
            t0 = time.time()
            list(pool.map(put, jobs))
            os.sync()
            dt = time.time() - t0
        print(f'[{where}] write   {len(jobs)} chunk files, {fbytes / 1e9:.2f} GB through Python open / write / close: {dt:6.2f} s  {fbytes / dt / 1e9:6.2f} GB/s (16 threads, incl. sync)', flush=True)
        shutil.rmtree(outdir, ignore_errors=True)
        # the same files by native threads (sq_write_files: what the stream writer uses since round 4), one call per level
        os.makedirs(outdir)
        t0 = time.time()
        for li, (buf, o) in enumerate(frames):
            keep = np.nonzero(np.diff(o) > 0)[0]
            if len(keep):
                native.write_files([os.path.join(outdir, f'{li}.{i}') for i in keep.tolist()], buf, np.concatenate([o[keep], o[keep[-1] + 1:keep[-1] + 2]]), 16)
        t1 = time.time()
        os.sync()
        print(f'[{where}] write   the same through sq_write_files: {t1 - t0:6.2f} s  {fbytes / (t1 - t0) / 1e9:6.2f} GB/s (16 native threads; + {time.time() - t1:.2f} s of sync)', flush=True)
        shutil.rmtree(outdir, ignore_errors=True)
        del region, planes, levels, encs, frames
        # ---- run: files -> store, streamed
        for rep in range(2):
            st.output_folder = os.path.join(tmp, f'out{rep}')
            with quiet():
                t0 = time.time()
                path = st.stream_region_to_zarr(0, st.regions[0])
                dt = time.time() - t0
            w, h = st.calculate_output_dimensions(0, st.regions[0])
            vox = st.num_c * st.num_z * w * h
            nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(path) for f in fs)
            print(f'[{where}] run     files -> {st.num_pyramid_levels}-level store (run {rep}): {dt:6.2f} s  {vox / dt / 1e9:5.2f} Gvoxel/s, store {nbytes / 1e9:.2f} GB', flush=True)
            shutil.rmtree(st.output_folder, ignore_errors=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


for where in (base, '/dev/shm'):
    if os.path.isdir(where):
        probe(where)

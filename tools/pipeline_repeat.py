"""The product path end to end, repeated: a multi-region, multi-timepoint acquisition on disk (tiles read by host
threads, staged, copied, registered, fused, pyramid levels, chunks encoded on the device and written by the writer
threads -- everything that runs on side streams and threads) through the CLI N times; every store's files are hashed
and compared with the first run's.  A race between the streams / threads shows as a run whose files differ."""
import hashlib, os, random, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import synth, stitcher_cli
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def digest_tree(root):
    h = hashlib.sha256()
    n = 0
    for d, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            p = os.path.join(d, f)
            rel = os.path.relpath(p, root)
            if rel.endswith(('.log',)) or 'shift_table' in rel or 'flatfield_info' in rel:
                continue
            with open(p, 'rb') as fh:
                data = fh.read()
            if f in ('.zattrs', '.zgroup', '.zarray') or f.endswith('.json'):
                continue      # metadata may carry paths / times; the voxels are in the chunk files
            h.update(rel.encode() + b'\0' + hashlib.sha256(data).digest())
            n += 1
    return h.hexdigest(), n


base = tempfile.mkdtemp(prefix='sq_pipe_', dir='/tmp')
try:
    spec = synth.GridSpec(rows=5, cols=4, tile_h=768, tile_w=1024, ov_y=96, ov_x=128, seed=21, channels=('Fluorescence_488_nm_Ex', 'Fluorescence_561_nm_Ex'),
                          nz=3, nt=2, regions=('A1', 'B2', 'C3'))
    acq = os.path.join(base, 'acq')
    t0 = time.perf_counter()
    synth.write_acquisition(spec, acq)
    print(f'acquisition written in {time.perf_counter() - t0:.1f} s: {spec.nt} timepoints x {len(spec.regions)} regions x {spec.rows}x{spec.cols} tiles x '
          f'{len(spec.channels)} ch x {spec.nz} z', flush=True)
    first = None
    differ = 0
    for it in range(N):
        for d in os.listdir(base):
            if d.startswith('acq_stitched_'):
                shutil.rmtree(os.path.join(base, d))
        random.seed(1234)      # the flatfield estimate samples tiles with the global generator, like the reference
        t0 = time.perf_counter()
        stitcher_cli.main(['-i', acq, '-r', '-ff', '--per-region-registration'] if it % 2 else ['-i', acq, '-r', '-ff'])
        out = [d for d in os.listdir(base) if d.startswith('acq_stitched_')]
        assert len(out) == 1, out
        dg, n = digest_tree(os.path.join(base, out[0]))
        key = it % 2
        if first is None:
            first = {}
        if key not in first:
            first[key] = dg
        elif dg != first[key]:
            differ += 1
            print(f'  run {it}: the store differs from the first run of its kind', flush=True)
        print(f'run {it} ({"per-region registration" if key else "one registration"}): {n} chunk files, digest {dg[:16]}, {time.perf_counter() - t0:.1f} s', flush=True)
    print(f'{N} runs: {differ} differ')
finally:
    shutil.rmtree(base, ignore_errors=True)

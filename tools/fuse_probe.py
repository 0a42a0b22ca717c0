"""Quick device probe: fusion throughput on a registered grid (not the bench contract)."""
import argparse
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, placement, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', type=int, default=16)
    ap.add_argument('--planes', type=int, default=8)
    ap.add_argument('--tile', type=int, default=2048)
    ap.add_argument('--ov', type=int, default=244)
    ap.add_argument('--flat', choices=['none', 'f32', 'f64'], default='none')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--nflats', type=int, default=1, help='distinct flatfields, dealt to planes in blocks (channels)')
    ap.add_argument('--drift', type=int, nargs=2, default=[3, -2])
    ap.add_argument('--check', action='store_true', help='compare plane 0 with the oracle')
    ap.add_argument('--dense', action='store_true', help='dense canvas stack (plane stride = Hc*Wc): no plane groups')
    ap.add_argument('--flags', type=int, default=0)
    ap.add_argument('--ab', type=int, default=None, help='also time these flags, alternating with --flags in the same process (same buffers)')
    ap.add_argument('--libs', default=None, help='comma-separated build variants (tools/build_variant.sh names, or "default") timed alternately in this process on the same buffers')
    ap.add_argument('--u8', action='store_true', help='uint8 tiles and canvas')
    ap.add_argument('--feather', action='store_true', help='feather plan (uncropped rectangles, blended overlaps)')
    ap.add_argument('--canvas-first', action='store_true', help='allocate the canvas before the tiles')
    ap.add_argument('--blocks', type=int, default=0, help='cap / set the launch grid (grid_blocks); with --flags 2 and a huge value: one workgroup per work unit')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    TDT, NDT, ESZ = (torch.uint8, 'uint8', 1) if a.u8 else (torch.uint16, 'uint16', 2)
    g, T = a.grid, a.tile
    shifts = placement.Shifts((a.drift[0], -a.ov), (-a.ov, a.drift[1]))
    rects = placement.grid_rects(g, g, T, T, shifts, crop=not a.feather)
    MODE = native.SQ_FUSE_FEATHER if a.feather else native.SQ_FUSE_OVERWRITE
    wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=shifts)
    t0 = time.time()
    plan = native.FusePlan(rects, T, T, hc, wc, MODE)
    print(f'plan: {plan.n_spans} spans, {plan.n_items} items, table {plan.table.nbytes/1e6:.2f} MB, '
          f'{time.time()-t0:.3f}s; canvas {hc}x{wc}; covered {plan.covered_voxels/(hc*wc):.3f}')
    spec = synth.GridSpec(rows=g, cols=g, tile_h=T, tile_w=T, ov_y=a.ov, ov_x=a.ov, seed=1)
    early = (native.empty_canvas(a.planes, hc, wc, TDT, dev) if not a.dense else
             torch.empty((a.planes, hc, wc), dtype=TDT, device=dev)) if a.canvas_first else None
    tiles = torch.empty((a.planes, g * g, T, T), dtype=TDT, device=dev)
    for p in range(a.planes):
        desc = np.zeros(g * g, dtype=native.SYNTH_DTYPE)
        for r in range(g):
            for c in range(g):
                oy, ox = spec.origin(r, c)
                desc[r * g + c] = (spec.scene_seed(0, 0, p, 0) % 2**64, spec.noise_seed(0, 0, p, 0, r * g + c) % 2**64, oy, ox)
        native.synth_tiles(desc, T, T, 200 if not a.u8 else 2, NDT, dev, out=tiles[p])
    torch.cuda.synchronize()
    canvas = early if early is not None else (native.empty_canvas(a.planes, hc, wc, TDT, dev) if not a.dense else
                                              torch.empty((a.planes, hc, wc), dtype=TDT, device=dev))
    print(f'tiles at {tiles.data_ptr():#x}, canvas at {canvas.data_ptr():#x}')
    flats = None
    if a.flat != 'none':
        g = synth.synthetic_flatfield(T, T, np.float32 if a.flat == 'f32' else np.float64)
        ffs = [torch.from_numpy(g).to(dev) for _ in range(a.nflats)]
        flats = [ffs[p * a.nflats // a.planes] for p in range(a.planes)]
    for _ in range(2):
        native.fuse_planes(plan, tiles, canvas, flats, flags=a.flags, grid_blocks=a.blocks)
    torch.cuda.synchronize()
    evs = []
    for _ in range(a.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.fuse_planes(plan, tiles, canvas, flats, flags=a.flags, grid_blocks=a.blocks)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = np.array([e0.elapsed_time(e1) for e0, e1 in evs])
    vox = a.planes * hc * wc
    alg = a.planes * (plan.covered_voxels * 2 * ESZ + (hc * wc - plan.covered_voxels) * ESZ)
    if a.feather:   # every tile pixel read once + the voxel written
        alg = a.planes * (a.grid * a.grid * T * T * 2 + hc * wc * 2)
    if a.check:
        from oracle import stitch_oracle as O
        fuse_o = (lambda t, r, h, w, f: O.fuse_plane_feather(t, r, h, w, f, out_dtype=np.uint16)) if a.feather else O.fuse_plane_overwrite
        want = fuse_o(list(tiles[0].cpu().numpy()), rects, hc, wc, None if flats is None else g)
        for p in sorted({0, a.planes - 1}):
            want_p = want if p == 0 else fuse_o(list(tiles[p].cpu().numpy()), rects, hc, wc, None if flats is None else g)
            print(f'plane {p} mismatched voxels vs oracle:', int(np.count_nonzero(canvas[p].cpu().numpy() != want_p)))
    print(f'fuse: {ms.mean():.3f} ms (min {ms.min():.3f}) -> {vox/ms.mean()/1e3:.1f} Mvoxel/s, '
          f'{alg/ms.mean()/1e6:.1f} GB/s algorithmic ({alg/ms.mean()/1e6/8000:.3f} of 8 TB/s)')
    if a.libs:
        import ctypes as C
        here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'image-stitcher_amd', 'csrc')
        handles, plans = {}, {}
        for name in a.libs.split(','):
            h = C.CDLL(os.path.join(here, 'libsquidstitch.so' if name == 'default' else f'libsquidstitch_{name}.so'))
            for fn, (res, args) in native.EXPORTS.items():
                getattr(h, fn).restype = res
                getattr(h, fn).argtypes = args
            handles[name] = h
            native._lib = h
            plans[name] = native.FusePlan(rects, T, T, hc, wc, MODE)
        for rnd in range(4):
            for name in handles:
                native._lib = handles[name]
                for _ in range(1 if rnd else 2):
                    native.fuse_planes(plans[name], tiles, canvas, flats, flags=a.flags, grid_blocks=a.blocks)
                evs = []
                for _ in range(a.steps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    native.fuse_planes(plans[name], tiles, canvas, flats, flags=a.flags, grid_blocks=a.blocks)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                m = np.array([e0.elapsed_time(e1) for e0, e1 in evs]).mean()
                bad = ''
                if a.check and rnd == 0:
                    bad = f', plane 0 mismatched voxels: {int(np.count_nonzero(canvas[0].cpu().numpy() != want))}'
                print(f'  round {rnd} {name}: {m:.3f} ms ({alg/m/1e6/8000:.4f} of 8 TB/s), {plans[name].n_items} items{bad}', flush=True)
    if a.ab is not None:
        for rnd in range(4):
            for fl in (a.flags, a.ab):
                evs = []
                for _ in range(a.steps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    native.fuse_planes(plan, tiles, canvas, flats, flags=fl, grid_blocks=a.blocks)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                m = np.array([e0.elapsed_time(e1) for e0, e1 in evs]).mean()
                print(f'  round {rnd} flags {fl}: {m:.3f} ms ({alg/m/1e6/8000:.4f} of 8 TB/s)', flush=True)


if __name__ == '__main__':
    main()

"""The feather kernels on config 3's geometry with the canvas in a mixed arena (what bench.py's feather leg measures), for A/B runs
of library variants (SQ_LIB_PATH) and experiment switches:  python tools/feather_probe.py [C=4] [Z=5] [reps=5]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from image_stitcher_amd import native, placement, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Z = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
g, T, P = 16, 2048, C * Z
dev = torch.device('cuda:0')
spec, truth, wc, hc, xs, ys, order, order_rc = bench.grid_setup(g, 3000)
need = native.canvas_bytes(P, hc, wc, torch.uint16)
# under rocprofv3 the card does not get released slices back: take exactly what the arena needs (bytes and times per launch of a
# counter pass are what matters there, not where the canvas lies)
profiled = any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) and bool(os.environ.get('FEATHER_PROBE_EXACT_ARENA', '1') != '0')
arena = native.DeviceArena(need, dev, candidate_bytes=need if profiled else None)
print('arena:', arena.info['class_slices'], arena.info['create_ms'], 'ms', flush=True)
rects = placement.grid_rects(g, g, T, T, truth, order=order_rc, crop=False)
plan = native.FusePlan(rects, T, T, hc, wc, native.SQ_FUSE_FEATHER)
rho = g * g * T * T / (hc * wc)
tiles = torch.empty((P, g * g, T, T), dtype=torch.uint16, device=dev)
for p in range(P):
    native.synth_tiles(bench.plane_desc(spec, g, p // Z, p % Z), T, T, spec.noise, 'uint16', dev, out=tiles[p])
ffs = [torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32) * np.float32(1 + 0.03125 * c)).to(dev) for c in range(C)]
plane_of_slot = [(s % C) * Z + s // C for s in range(P)]
flats = [ffs[p // Z] for p in plane_of_slot]
ptrs = bench.tile_pointer_table(tiles, plane_of_slot, torch.tensor(order, dtype=torch.int64), dev)
print(f'{P} planes, canvas {hc}x{wc}, feather plan {plan.n_items} items, rho {rho:.4f}; lib {native.LIB_PATH}', flush=True)


if os.environ.get('FEATHER_COVER'):      # how many canvas voxels 0 / 1 / 2 / 3+ tiles cover
    cover = torch.zeros((hc, wc), dtype=torch.int8, device=dev)
    for top, left, h, w, y, x in rects:
        cover[y:y + h, x:x + w] += 1
    cnt = torch.bincount(cover.view(-1).to(torch.int64), minlength=5).cpu().numpy()
    print('canvas voxels by number of covering tiles:', {k: f'{c / (hc * wc):.4f}' for k, c in enumerate(cnt)}, flush=True)
    del cover


def run(cv, n, fl, out_bytes, name):
    fp = native.pointer_table(fl, dev) if fl else None
    ms = []
    for k in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.fuse_planes(plan, None, cv, fl, tile_ptrs=ptrs[:n * g * g], flat_ptrs=fp)
        e1.record()
        torch.cuda.synchronize()
        if k >= 2:
            ms.append(e0.elapsed_time(e1))
    m = float(np.mean(ms))
    alg = n * (hc * wc * (2 * rho + out_bytes) + (T * T * 4 if fl else 0))
    print(f'{name:34s} {n:3d} planes  {m:8.3f} ms  {alg / m / 1e6 / 8000:.4f} of 8 TB/s  (digest {bench.plane_digest(cv[0].view(torch.uint16)[:, :wc]) % 100000})', flush=True)


for rnd in range(2):
    arena.reset()
    cv = native.empty_canvas(P, hc, wc, torch.uint16, dev, arena=arena)
    run(cv, P, flats, 2, 'uint16 canvas, float32 gains')
    run(cv, P, None, 2, 'uint16 canvas, no gains')
    del cv
    arena.reset()
    cf = native.empty_canvas(P // 2, hc, wc, torch.float32, dev, arena=arena)
    run(cf, P // 2, flats[:P // 2], 4, 'float32 canvas, float32 gains')
    run(cf, P // 2, None, 4, 'float32 canvas, no gains')
    del cf

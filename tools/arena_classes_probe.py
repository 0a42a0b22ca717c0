"""Two halves against three thirds: the REAL fusion kernel (config-3 geometry, 20 planes) with its canvas in an arena mapped over all
classes of the card against one mapped over the two largest only (SQ_ARENA_TWO_CLASSES), and in a plain allocation; one process, alternating.
    python tools/arena_classes_probe.py [planes=20] [rounds=3]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from image_stitcher_amd import native, placement, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g, T, C = 16, 2048, 4
Z = P // C
dev = torch.device('cuda:0')
spec, truth, wc, hc, xs, ys, order, order_rc = bench.grid_setup(g, 3000)
rects = placement.grid_rects(g, g, T, T, truth, order=order_rc)
plan = native.FusePlan(rects, T, T, hc, wc, native.SQ_FUSE_OVERWRITE)
need = native.canvas_bytes(P, hc, wc, torch.uint16)
arenas = {'all classes': native.DeviceArena(need, dev), 'two classes': native.DeviceArena(need, dev, two_classes=True)}
for k, a in arenas.items():
    print(f'{k}: class_slices {a.info["class_slices"]} of candidates {a.info["class_candidates"]}, {a.info["create_ms"]:.0f} ms', flush=True)
canvases = {k: native.empty_canvas(P, hc, wc, torch.uint16, dev, arena=a) for k, a in arenas.items()}
if P <= 20:      # (more planes: the card has no room for a third canvas beside the tiles)
    canvases['plain allocation'] = native.empty_canvas(P, hc, wc, torch.uint16, dev)
tiles = torch.empty((P, g * g, T, T), dtype=torch.uint16, device=dev)
for p in range(P):
    native.synth_tiles(bench.plane_desc(spec, g, p // Z, p % Z), T, T, spec.noise, 'uint16', dev, out=tiles[p])
ffs = [torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32) * np.float32(1 + 0.03125 * c)).to(dev) for c in range(C)]
plane_of_slot = [(s % C) * Z + s // C for s in range(P)]
flats = [ffs[p // Z] for p in plane_of_slot]
fp = native.pointer_table(flats, dev)
ptrs = bench.tile_pointer_table(tiles, plane_of_slot, torch.tensor(order, dtype=torch.int64), dev)
alg = bench.algorithmic_bytes(P, plan.covered_voxels, hc, wc, True)
for r in range(rounds):
    for k, cv in canvases.items():
        ms = []
        for i in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.fuse_planes(plan, None, cv, flats, tile_ptrs=ptrs, flat_ptrs=fp)
            e1.record()
            torch.cuda.synchronize()
            if i:
                ms.append(e0.elapsed_time(e1))
        m = float(np.mean(ms))
        print(f'round {r}  {k:18s} {m:8.3f} ms  {alg / m / 1e6 / 8000:.4f} of 8 TB/s  (digest {bench.plane_digest(cv[0]) % 100000})', flush=True)

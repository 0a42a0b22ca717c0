"""Does the fusion rate depend on WHERE the canvas (and the tiles) sit in memory?  One process, one tile stack, the canvas
re-allocated at different offsets inside a big slab (and at fresh allocations); config-3 geometry, float32 gains, 20 planes."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, placement, synth

dev = torch.device('cuda:0')
g, T, OV, P = 16, 2048, 244, 20
shifts = placement.Shifts((3, -OV), (-OV, -2))
rects = placement.grid_rects(g, g, T, T, shifts)
wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=shifts)
plan = native.FusePlan(rects, T, T, hc, wc)
spec = synth.GridSpec(rows=g, cols=g, tile_h=T, tile_w=T, ov_y=OV, ov_x=OV, seed=1)
tiles = torch.empty((P, g * g, T, T), dtype=torch.uint16, device=dev)
for p in range(P):
    desc = np.zeros(g * g, dtype=native.SYNTH_DTYPE)
    for r in range(g):
        for c in range(g):
            oy, ox = spec.origin(r, c)
            desc[r * g + c] = (spec.scene_seed(0, 0, p, 0) % 2**64, spec.noise_seed(0, 0, p, 0, r * g + c) % 2**64, oy, ox)
    native.synth_tiles(desc, T, T, 200, 'uint16', dev, out=tiles[p])
ff = [torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32)).to(dev) for _ in range(2)]
flats = [ff[p * 2 // P] for p in range(P)]
stride = -(-(hc * wc) // 64) * 64
alg = P * (plan.covered_voxels * 4 + (hc * wc - plan.covered_voxels) * 2)


def rate(canvas):
    for _ in range(2):
        native.fuse_planes(plan, tiles, canvas, flats)
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); native.fuse_planes(plan, tiles, canvas, flats); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return alg / min(ms) / 1e6 / 8000


print(f'tiles at {tiles.data_ptr():#x}')
slab = torch.empty(P * stride + (64 << 20), dtype=torch.uint16, device=dev)
for off_bytes in (0, 128, 4096, 65536, 1 << 20, (1 << 21) + 4096, 3 << 20, 16 << 20, (32 << 20) + 128 * 77):
    canvas = slab[off_bytes // 2:].as_strided((P, hc, wc), (stride, wc, 1))
    print(f'canvas at slab + {off_bytes:>9d} ({canvas.data_ptr():#x}): {rate(canvas):.3f}', flush=True)
# other plane strides: + one 4 KiB page, + an odd number of 128-byte lines per plane
for extra in (2048, 64 * 33, 64 * 1025):
    st = stride + extra
    big = torch.empty(P * st, dtype=torch.uint16, device=dev)
    canvas = big.as_strided((P, hc, wc), (st, wc, 1))
    print(f'plane stride + {extra * 2} bytes ({canvas.data_ptr():#x}): {rate(canvas):.3f}', flush=True)
    del big, canvas
keep = []
for k in range(4):
    keep.append(torch.empty((k + 1) * (37 << 20), dtype=torch.uint8, device=dev))     # push the next allocation somewhere else
    canvas = native.empty_canvas(P, hc, wc, torch.uint16, dev)
    print(f'fresh canvas {k} at {canvas.data_ptr():#x}: {rate(canvas):.3f}', flush=True)
    keep.append(canvas)

"""The hot path at full size, repeated: config 3's geometry (16x16 grid of 2048^2 tiles, float32 gains, plane groups,
device queues), 20 resident planes, N launches into a poisoned canvas, every launch compared on the device with the first
one -- whose planes 0 and 10 are compared with the oracle.  A lost or misplaced work item anywhere in 21 Gvoxel shows."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, placement, synth
from oracle import stitch_oracle as O
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g, T, P = 16, 2048, 20
spec = synth.GridSpec(rows=g, cols=g, tile_h=T, tile_w=T, ov_y=244, ov_x=244, seed=5)
tiles = torch.empty((P, g * g, T, T), dtype=torch.uint16, device=dev)
for p in range(P):
    desc = np.zeros(g * g, dtype=native.SYNTH_DTYPE)
    for r in range(g):
        for c in range(g):
            oy, ox = spec.origin(r, c)
            desc[r * g + c] = (spec.scene_seed(0, 0, p, 0) % 2**64, spec.noise_seed(0, 0, p, 0, r * g + c) % 2**64, oy, ox)
    native.synth_tiles(desc, T, T, 200, 'uint16', dev, out=tiles[p])
sh = placement.Shifts((3, -244), (-244, -2))
wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=sh)
rects = placement.grid_rects(g, g, T, T, sh)
plan = native.FusePlan(rects, T, T, hc, wc, expand_on_device=True)
ffs = [torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32) * np.float32(1 + 0.03125 * c)).to(dev) for c in range(2)]
flats = [ffs[p // 10] for p in range(P)]
ref = native.empty_canvas(P, hc, wc, torch.uint16, dev)
native.fuse_planes(plan, tiles, ref, flats)
torch.cuda.synchronize()
for p in (0, 10):
    t0 = time.perf_counter()
    want = O.fuse_plane_overwrite([tiles[p, i].cpu().numpy() for i in range(g * g)], rects, hc, wc, flats[p].cpu().numpy())
    bad = int(np.count_nonzero(ref[p].cpu().numpy() != want))
    print(f'plane {p} of the first launch against the oracle: {bad} voxels differ ({time.perf_counter() - t0:.0f} s)', flush=True)
    assert bad == 0
canvas = native.empty_canvas(P, hc, wc, torch.uint16, dev)
differ = 0
t0 = time.perf_counter()
for it in range(N):
    canvas.view(torch.int16).fill_(-1 - (it % 5))
    native.fuse_planes(plan, tiles, canvas, flats)
    n = sum(int(not torch.equal(canvas[p], ref[p])) for p in range(P))
    if n:
        differ += 1
        print(f'  launch {it}: {n} planes differ', flush=True)
print(f'{N} launches of {P} planes ({P * hc * wc / 1e9:.1f} Gvoxel each): {differ} differ from the first ({time.perf_counter() - t0:.1f} s)')

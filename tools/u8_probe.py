"""uint8 planes through the plane groups (round 4) against one plane at a time, alternated in ONE process on the same buffers: config
3's geometry with uint8 tiles, C gain images x Z planes, canvas in a DeviceArena.   python tools/u8_probe.py [C=4] [Z=5] [rounds=3]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from image_stitcher_amd import native, placement, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Z = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g, T, P = 16, 2048, C * Z
dev = torch.device('cuda:0')
spec, truth, wc, hc, xs, ys, order, order_rc = bench.grid_setup(g, 3000)
arena = native.DeviceArena(native.canvas_bytes(P, hc, wc, torch.uint8), dev)
canvas = native.empty_canvas(P, hc, wc, torch.uint8, dev, arena=arena)
rects = placement.grid_rects(g, g, T, T, truth, order=order_rc)
plan = native.FusePlan(rects, T, T, hc, wc, native.SQ_FUSE_OVERWRITE)
tiles = torch.empty((P, g * g, T, T), dtype=torch.uint8, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
for p in range(P):
    tiles[p].copy_(torch.randint(0, 256, (g * g, T, T), dtype=torch.uint8, device=dev, generator=gen))
ffs = [torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32) * np.float32(1 + 0.03125 * c)).to(dev) for c in range(C)]
plane_of_slot = [(s % C) * Z + s // C for s in range(P)]
flats = [ffs[p // Z] for p in plane_of_slot]
fp = native.pointer_table(flats, dev)
esz = T * T
ptrs = (tiles.data_ptr() + torch.tensor(plane_of_slot, dtype=torch.int64)[:, None] * (tiles.stride(0)) + torch.tensor(order, dtype=torch.int64)[None, :] * esz).reshape(-1).to(dev)
covered = plan.covered_voxels
alg = P * (covered * 2 + (hc * wc - covered) * 1 + T * T * 4)
print(f'{P} uint8 planes ({C} gain images x {Z}), canvas {hc}x{wc}, arena {arena.info["class_slices"]}; algorithmic bytes per launch {alg / 1e9:.1f} GB', flush=True)


def run(flags, gains=True):
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.fuse_planes(plan, None, canvas, flats if gains else None, tile_ptrs=ptrs, flat_ptrs=fp if gains else None, flags=flags)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


want = None
for r in range(rounds):
    for name, flags in (('one plane at a time (rounds 1-3)', native.SQ_FUSE_NO_PLANE_GROUPS), ('plane groups', 0)):
        ms = run(flags)
        d = [int(canvas[s].to(torch.int64).sum().item()) for s in (0, P - 1)]
        want = want or d
        assert d == want, 'the variants fuse different voxels'
        print(f'round {r}: float32 gains, {name:34s} {ms:7.3f} ms  {alg / ms / 1e6 / 8000:.4f} of 8 TB/s', flush=True)
    algn = alg - P * T * T * 4
    for name, flags in (('one plane at a time (rounds 1-3)', native.SQ_FUSE_NO_PLANE_GROUPS), ('plane groups', 0)):
        ms = run(flags, gains=False)
        print(f'round {r}: no gains,      {name:34s} {ms:7.3f} ms  {algn / ms / 1e6 / 8000:.4f} of 8 TB/s', flush=True)

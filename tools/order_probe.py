"""Where a group's planes lie in the canvas allocation, on the REAL fusion kernel (fuse_overwrite_zg_kernel), alternated
in ONE process on the SAME buffers:
    slot order 'plane'  : plane p = c * Z + z at canvas slot p (round 2)        | groups 'consecutive' (round 2: the first five
    slot order 'spread' : slot s holds channel s % C, z = s // C                | planes of a key, the next five ...) or 'dealt'
                                                                                | (round-robin over the key's groups, the default)
    python tools/order_probe.py [grid=16] [C=4] [Z=10] [rounds=3]
(tools/membw_gains found stretches of tens of GiB of device memory to behave like separate resources for the row-segment
write pattern: a group whose planes all lie in one stretch writes at 0.56 of peak, one spread over two at 0.74.)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from image_stitcher_amd import native, placement, synth

g = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
Z = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
nogain = len(sys.argv) > 5 and sys.argv[5] == 'nogain'     # planes WITHOUT a flatfield: grouped (round 3) against plane by plane
P, T = C * Z, 2048
dev = torch.device('cuda:0')
spec, truth, wc, hc, xs, ys, order, order_rc = bench.grid_setup(g, 3000)
rects = placement.grid_rects(g, g, T, T, truth, order=order_rc)
plan = native.FusePlan(rects, T, T, hc, wc, native.SQ_FUSE_OVERWRITE)
# SQ_PROBE_LAYOUT=interleaved: ONE arena [canvas slot 0 | tile stack 0 | canvas slot 1 | tile stack 1 | ...] -- the canvas
# planes then lie (canvas plane + tile stack) bytes apart, over all the memory the job holds, instead of side by side
interleaved = os.environ.get('SQ_PROBE_LAYOUT') == 'interleaved'
if interleaved:
    cplane = -(-(hc * wc * 2) // 4096) * 4096
    tplane = g * g * T * T * 2
    arena = torch.empty(P * (cplane + tplane), dtype=torch.uint8, device=dev)
    canvas = arena.view(torch.uint16).as_strided((P, hc, wc), ((cplane + tplane) // 2, wc, 1))
    tiles = arena.view(torch.uint16).as_strided((P, g * g, T, T), ((cplane + tplane) // 2, T * T, T, 1), storage_offset=cplane // 2)
else:
    tiles = torch.empty((P, g * g, T, T), dtype=torch.uint16, device=dev)
    canvas = native.empty_canvas(P, hc, wc, torch.uint16, dev)
for p in range(P):
    native.synth_tiles(bench.plane_desc(spec, g, p // Z, p % Z), T, T, spec.noise, 'uint16', dev, out=tiles[p])
ffs = [torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32) * np.float32(1 + 0.03125 * c)).to(dev) for c in range(C)]
alg = bench.algorithmic_bytes(P, plan.covered_voxels, hc, wc, not nogain)
tile_order = torch.tensor(order, dtype=torch.int64)
print(f'{P} planes ({C} gain images x {Z}) of the {g}x{g} grid, canvas {hc}x{wc}: canvas planes {canvas.stride(0) * 2 / 2**30:.2f} GiB apart '
      f'({"one arena, canvas slots and tile stacks interleaved" if interleaved else "canvas and tiles in two allocations"}), '
      f'{tiles.numel() * 2 / 2**30:.1f} GiB of tiles', flush=True)


def setup(slot_order):
    plane_of_slot = [(s % C) * Z + s // C for s in range(P)] if slot_order == 'spread' else list(range(P))
    flats = [ffs[p // Z] for p in plane_of_slot]
    ptrs = (tiles.data_ptr() + torch.tensor(plane_of_slot, dtype=torch.int64)[:, None] * (tiles.stride(0) * 2) + tile_order[None, :] * (T * T * 2))
    return plane_of_slot, flats, native.pointer_table(flats, dev), ptrs.reshape(-1).to(dev)


def run(cfg, flags, reps=3):
    plane_of_slot, flats, fp, ptrs = cfg
    best = 1e9
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.fuse_planes(plan, None, canvas, None if nogain else flats, tile_ptrs=ptrs, flat_ptrs=None if nogain else fp, flags=flags)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def digest(cfg):
    """order-independent check: digest of every PLANE (not slot)"""
    plane_of_slot = cfg[0]
    return {p: bench.plane_digest(canvas[s]) for s, p in enumerate(plane_of_slot) if p in (0, Z - 1, P - 1)}


cfgs = {name: setup(name) for name in ('plane', 'spread')}
want = None
for r in range(rounds):
    for name in ('plane', 'spread'):
        extra = ()
        for gname, flags in ((('one plane at a time', native.SQ_FUSE_NO_PLANE_GROUPS),) if nogain else ()) + (('consecutive', native.SQ_FUSE_CONSECUTIVE_GROUPS), ('dealt', 0)) + extra:
            ms = run(cfgs[name], flags)
            d = digest(cfgs[name])
            want = want or d
            assert d == want, 'the fused planes differ between the variants'
            print(f'round {r}: slots {name:6s} groups {gname:28s}  {ms:7.3f} ms   {alg / ms / 1e6 / 8000:.4f} of 8 TB/s', flush=True)
print('every variant produced the same planes (digests of planes 0, Z-1, P-1)')

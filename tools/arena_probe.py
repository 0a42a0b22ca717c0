"""The canvas in a MIXED arena (native.DeviceArena: physical slices classified and mapped round-robin over the card's memory
classes, csrc/arena.hip) against a plain allocation, on the REAL fusion kernel, alternated in ONE process with the same tiles:
    canvas 'plain' : native.empty_canvas (one torch allocation)         x slot order 'plane' / 'spread'  x groups consecutive / dealt
    canvas 'mixed' : native.empty_canvas(arena=DeviceArena)              x the same
    python tools/arena_probe.py [grid=16] [C=4] [Z=6] [rounds=3] [nogain|feather]
Also checks that a tensor carved from the arena aliases it (writes through torch are seen by the kernel's reads and back)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from image_stitcher_amd import native, placement, synth

g = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
Z = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
variant = sys.argv[5] if len(sys.argv) > 5 else ''
nogain = variant == 'nogain'
P, T = C * Z, 2048
dev = torch.device('cuda:0')
spec, truth, wc, hc, xs, ys, order, order_rc = bench.grid_setup(g, 3000)
rects = placement.grid_rects(g, g, T, T, truth, order=order_rc)
plan = native.FusePlan(rects, T, T, hc, wc, native.SQ_FUSE_OVERWRITE)

need = native.canvas_bytes(P, hc, wc, torch.uint16)
arena = native.DeviceArena(need + (64 << 20), dev)
print('arena:', arena.info, flush=True)
# aliasing check: a tensor taken from the arena is the arena's memory
t = arena.take(1 << 20)
t.copy_(torch.arange(1 << 20, dtype=torch.int64, device=dev).to(torch.uint8))
u = torch.as_tensor(native._RawDeviceMemory(arena, arena.base, 1 << 20), device=dev)
assert torch.equal(t, u) and t.data_ptr() == arena.base and int(t[12345].item()) == 12345 % 256
v = t.view(torch.uint16)
v[7] = 0xBEEF
assert int(u[14].item()) == 0xEF and int(u[15].item()) == 0xBE
assert t.cpu().numpy()[14] == 0xEF
arena.reset()
canvases = {'plain': native.empty_canvas(P, hc, wc, torch.uint16, dev), 'mixed': native.empty_canvas(P, hc, wc, torch.uint16, dev, arena=arena)}
assert canvases['mixed'].data_ptr() == arena.base and canvases['mixed'].stride() == canvases['plain'].stride()
tiles = torch.empty((P, g * g, T, T), dtype=torch.uint16, device=dev)
for p in range(P):
    native.synth_tiles(bench.plane_desc(spec, g, p // Z, p % Z), T, T, spec.noise, 'uint16', dev, out=tiles[p])
ffs = [torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32) * np.float32(1 + 0.03125 * c)).to(dev) for c in range(C)]
alg = bench.algorithmic_bytes(P, plan.covered_voxels, hc, wc, not nogain)
tile_order = torch.tensor(order, dtype=torch.int64)
print(f'{P} planes ({C} gain images x {Z}) of the {g}x{g} grid, canvas {hc}x{wc} ({canvases["plain"].stride(0) * 2 / 2**30:.2f} GiB per plane), '
      f'{tiles.numel() * 2 / 2**30:.1f} GiB of tiles in a plain allocation', flush=True)


def setup(slot_order):
    plane_of_slot = [(s % C) * Z + s // C for s in range(P)] if slot_order == 'spread' else list(range(P))
    flats = [ffs[p // Z] for p in plane_of_slot]
    ptrs = (tiles.data_ptr() + torch.tensor(plane_of_slot, dtype=torch.int64)[:, None] * (tiles.stride(0) * 2) + tile_order[None, :] * (T * T * 2))
    return plane_of_slot, flats, native.pointer_table(flats, dev), ptrs.reshape(-1).to(dev)


def run(canvas, cfg, flags, reps=3):
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


def digest(canvas, cfg):
    return {p: bench.plane_digest(canvas[s]) for s, p in enumerate(cfg[0]) if p in (0, Z - 1, P - 1)}


cfgs = {name: setup(name) for name in ('plane', 'spread')}
want = None
for r in range(rounds):
    for cname in ('plain', 'mixed'):
        for name in ('plane', 'spread'):
            for gname, flags in (('consecutive', native.SQ_FUSE_CONSECUTIVE_GROUPS), ('dealt', 0)):
                ms = run(canvases[cname], cfgs[name], flags)
                d = digest(canvases[cname], cfgs[name])
                want = want or d
                assert d == want, 'the fused planes differ between the variants'
                print(f'round {r}: canvas {cname:5s} slots {name:6s} groups {gname:11s}  {ms:7.3f} ms   {alg / ms / 1e6 / 8000:.4f} of 8 TB/s', flush=True)
print('every variant produced the same planes (digests of planes 0, Z-1, P-1)')

"""Repeatability stress: every kernel family launched a few hundred times on the same inputs, every result compared bit
for bit with the first one (and the fusion with the oracle): a data race shows as a launch that differs.  Written after
round 3 found the device queues losing work in 1-2 % of small launches (tools/queue_stress.py)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, placement, registration
from oracle import stitch_oracle as O
dev = torch.device('cuda:0')
rng = np.random.default_rng(11)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def report(name, bad, reps, t0):
    print(f'{name}: {bad} of {reps} launches differ from the first ({time.perf_counter() - t0:.1f} s)', flush=True)


# --- registration: power-of-two, mixed-radix, Bluestein crops; a batch and a single pair ------------------------
T = 2048
tiles = torch.randint(0, 65535, (12, T, T), dtype=torch.int32, device=dev).to(torch.uint16)
mm = native.tile_minmax(tiles)
for n0, n1, npairs in ((1024, 256, 24), (256, 1024, 24), (1500, 300, 12), (300, 1500, 12), (2038, 312, 6), (312, 2038, 6), (2002, 300, 6), (1024, 256, 1), (63, 77, 5)):
    t0 = time.perf_counter()
    src = tiles
    pairs = np.zeros(npairs, dtype=native.PAIR_DTYPE)
    pairs['ref_tile'] = np.arange(npairs) % 12
    pairs['mov_tile'] = (np.arange(npairs) + 1) % 12
    pairs['ref_y0'], pairs['ref_x0'], pairs['mov_y0'], pairs['mov_x0'] = 3, 5, 0, 0
    first = native.register_pairs(src, mm, pairs, n0, n1, 10, native.SQ_NORM_PHASE).tobytes()
    reps = max(20, REPS // 3)
    bad = sum(native.register_pairs(src, mm, pairs, n0, n1, 10, native.SQ_NORM_PHASE).tobytes() != first for _ in range(reps))
    report(f'registration {npairs} pairs of {n0}x{n1}', bad, reps, t0)

# --- plan expansion on the device ---------------------------------------------------------------------------------
t0 = time.perf_counter()
g, TT = 16, 2048
sh = placement.Shifts((3, -244), (-244, -2))
wc, hc = placement.canvas_size(g, g, TT, TT, use_registration=True, shifts=sh)
rects = placement.grid_rects(g, g, TT, TT, sh)
host = torch.from_numpy(native.FusePlan(rects, TT, TT, hc, wc).table.copy()).to(dev)
bad = 0
for _ in range(REPS):
    p = native.FusePlan(rects, TT, TT, hc, wc, expand_on_device=True)
    bad += int(not torch.equal(p.device_table(dev), host))
report('plan expansion (16x16 grid, 79 508 items) against the host table', bad, REPS, t0)

# --- fusion: the plane-group kernel with queues and gains on a small registered grid, against the oracle ----------
t0 = time.perf_counter()
g, th = 6, 192
sh = placement.Shifts((2, -30), (-28, -3))
wc, hc = placement.canvas_size(g, g, th, th, use_registration=True, shifts=sh)
rects = placement.grid_rects(g, g, th, th, sh)
planes = 10
tl = rng.integers(0, 65536, size=(planes, g * g, th, th)).astype(np.uint16)
flat = (0.5 + rng.random((th, th))).astype(np.float32)
want = torch.from_numpy(np.stack([O.fuse_plane_overwrite(list(tl[p]), rects, hc, wc, flat) for p in range(planes)]).astype(np.int32)).to(dev)
d_tl, d_flat = torch.from_numpy(tl).to(dev), torch.from_numpy(flat).to(dev)
plan = native.FusePlan(rects, th, th, hc, wc)
for name, flags in (('plane groups, queues', native.SQ_FUSE_FORCE_QUEUES), ('per-plane, queues', native.SQ_FUSE_FORCE_QUEUES | native.SQ_FUSE_NO_PLANE_GROUPS),
                    ('plane groups, static', native.SQ_FUSE_FORCE_STATIC)):
    bad = 0
    for _ in range(REPS):
        canvas = native.empty_canvas(planes, hc, wc, torch.uint16, dev)
        canvas.view(torch.int16).fill_(-7)
        native.fuse_planes(plan, d_tl, canvas, [d_flat] * planes, flags=flags)
        bad += int((canvas.to(torch.int32) != want).sum() > 0)
    report(f'overwrite fusion with gains, {name}, against the oracle', bad, REPS, t0)

# --- the same over other block / unit ratios: tiny and mid-size grids, plane counts that leave partial groups, no gains / f64 gains
for g2, th2, planes2, gkind in ((2, 64, 37, 'f32'), (3, 96, 5, 'none'), (8, 128, 13, 'f64'), (5, 160, 24, 'f32'), (1, 256, 11, 'none')):
    t0 = time.perf_counter()
    sh2 = placement.Shifts((2, -(th2 // 8)), (-(th2 // 9), -3)) if g2 > 1 else placement.Shifts()
    wc2, hc2 = placement.canvas_size(g2, g2, th2, th2, use_registration=g2 > 1, shifts=sh2) if g2 > 1 else (th2, th2)
    rects2 = placement.grid_rects(g2, g2, th2, th2, sh2) if g2 > 1 else np.array([(0, 0, th2, th2, 0, 0)])
    tl2 = rng.integers(0, 65536, size=(planes2, g2 * g2, th2, th2)).astype(np.uint16)
    fl2 = None if gkind == 'none' else (0.5 + rng.random((th2, th2))).astype(np.float32 if gkind == 'f32' else np.float64)
    want2 = torch.from_numpy(np.stack([O.fuse_plane_overwrite(list(tl2[p]), rects2, hc2, wc2, fl2) for p in range(planes2)]).astype(np.int32)).to(dev)
    d_tl2 = torch.from_numpy(tl2).to(dev)
    d_fl2 = None if fl2 is None else [torch.from_numpy(fl2).to(dev)] * planes2
    plan2 = native.FusePlan(rects2, th2, th2, hc2, wc2)
    bad = 0
    for _ in range(REPS):
        canvas = native.empty_canvas(planes2, hc2, wc2, torch.uint16, dev)
        canvas.view(torch.int16).fill_(-7)
        native.fuse_planes(plan2, d_tl2, canvas, d_fl2, flags=native.SQ_FUSE_FORCE_QUEUES)
        bad += int((canvas.to(torch.int32) != want2).sum() > 0)
    report(f'overwrite fusion {g2}x{g2} grid of {th2}^2 tiles, {planes2} planes, gains {gkind}, plane groups + queues, against the oracle', bad, REPS, t0)

# --- pyramid level and chunk encoder -------------------------------------------------------------------------------
t0 = time.perf_counter()
img = torch.from_numpy((2000 + (np.add.outer(np.arange(1500) * 3, np.arange(1900) * 2) // 4 + rng.integers(0, 8, (1500, 1900))) % 60000).astype(np.uint16)[None].repeat(2, 0)).to(dev)
first = native.downsample2(img).clone()
bad = sum(int(not torch.equal(native.downsample2(img), first)) for _ in range(REPS))
report('pyramid level 1500x1900 x 2 planes', bad, REPS, t0)
t0 = time.perf_counter()
b = native.blosc_encode_planes(img, 512, 512)
torch.cuda.synchronize()
off0, out0 = b.offsets.clone(), b.out[:int(b.offsets[-1])].clone()
bad = 0
for _ in range(max(20, REPS // 3)):
    b = native.blosc_encode_planes(img, 512, 512, buffers=b)
    torch.cuda.synchronize()
    bad += int(not (torch.equal(b.offsets, off0) and torch.equal(b.out[:int(b.offsets[-1])], out0)))
report('Blosc chunk encoder, 24 chunks', bad, max(20, REPS // 3), t0)

# --- flatfield estimate (BaSiC restatement: dozens of kernel launches per fit), min/max, feather plane groups with gains ---
t0 = time.perf_counter()
stack = torch.from_numpy((3000 + 800 * np.sin(np.add.outer(np.arange(512), np.arange(640)) / 97.0)[None] * (0.5 + rng.random((24, 1, 1)))
                          + rng.integers(0, 200, (24, 512, 640))).astype(np.uint16)).to(dev)
first, info0 = native.basic_fit(stack)
first = first.clone()
reps = max(10, REPS // 10)
bad = 0
for _ in range(reps):
    out, info = native.basic_fit(stack)
    bad += int(not (torch.equal(out, first) and info == info0))
report(f'BaSiC flatfield fit of 24 tiles of 512x640 ({info0["reweight_iterations"]} reweightings, {info0["ladmap_iterations"]} inner iterations)', bad, reps, t0)
t0 = time.perf_counter()
first = native.tile_minmax(tiles).clone()
bad = sum(int(not torch.equal(native.tile_minmax(tiles), first)) for _ in range(REPS))
report('tile min / max of 12 tiles of 2048^2', bad, REPS, t0)
t0 = time.perf_counter()
planf = native.FusePlan(rects, th, th, hc, wc, native.SQ_FUSE_FEATHER)
wantf = torch.from_numpy(np.stack([O.fuse_plane_feather(list(tl[p]), rects, hc, wc, flat, out_dtype=np.uint16) for p in range(planes)]).astype(np.int32)).to(dev)
bad = 0
for _ in range(REPS):
    canvas = native.empty_canvas(planes, hc, wc, torch.uint16, dev)
    canvas.view(torch.int16).fill_(-7)
    native.fuse_planes(planf, d_tl, canvas, [d_flat] * planes, flags=native.SQ_FUSE_FORCE_QUEUES)
    bad += int((canvas.to(torch.int32) != wantf).sum() > 0)
report('feather fusion with gains, plane groups + queues, against the oracle', bad, REPS, t0)

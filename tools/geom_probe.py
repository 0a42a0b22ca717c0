"""Which part of the grid geometry costs bandwidth?  Synthetic rect sets, no flatfield."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native
dev = torch.device('cuda:0')
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    return min(x.elapsed_time(y) for x, y in evs)
G, T, P = 16, 2048, 4
tiles = torch.empty((P, G * G, T, T), dtype=torch.uint16, device=dev); tiles.fill_(9)
def run(name, crop_l, crop_t, step_x, step_y, pad_w=0, pad_h=0, off=0):
    rects = []
    for r in range(G):
        for c in range(G):
            rects.append((crop_t, crop_l, T - 2 * crop_t, T - 2 * crop_l, off + r * step_y, off + c * step_x))
    wc = off + (G - 1) * step_x + T - 2 * crop_l + pad_w
    hc = off + (G - 1) * step_y + T - 2 * crop_t + pad_h
    plan = native.FusePlan(np.array(rects), T, T, hc, wc)
    canvas = torch.empty((P, hc, wc), dtype=torch.uint16, device=dev)
    ms = timeit(lambda: native.fuse_planes(plan, tiles, canvas))
    cov = plan.covered_voxels
    alg = P * (cov * 4 + (hc * wc - cov) * 2)
    print(f'{name:58s} canvas {hc}x{wc} cov {cov/(hc*wc):.3f} spans {plan.n_spans:5d}: {ms:7.3f} ms {alg/ms/1e6:7.1f} GB/s', flush=True)
    del canvas
run('full tiles, abutting', 0, 0, T, T)
run('crop 256 horizontally only (w 1536 = 3 full wave-instrs)', 256, 0, T - 512, T)
run('crop 192 horizontally only (w 1664 = 3.25 wave-instrs)', 192, 0, T - 384, T)
run('crop 128 (w 1792), abutting, aligned', 128, 128, T - 256, T - 256)
run('crop 120 (w 1808), abutting, aligned', 120, 120, T - 240, T - 240)
run('crop 122 (w 1804), abutting (dst phase varies)', 122, 122, T - 244, T - 244)
run('crop 122, abutting, canvas +3660 zero rows', 122, 122, T - 244, T - 244, pad_h=3660)
run('crop 122, 3-px gaps between tiles (thin zero spans)', 122, 122, T - 244 + 3, T - 244 + 3)
run('full tiles, abutting, canvas width +5 (odd pitch)', 0, 0, T, T, pad_w=5)
run('crop 128 horizontally only', 128, 0, T - 256, T)
run('crop 128 vertically only', 0, 128, T, T - 256)


def run_src(name, src_x, w=1792):
    """Destination seams on line boundaries (w = 1792 px = 28 lines, pitch a multiple of 64 px); only the
    SOURCE phase varies: how much do tile rows read at an odd byte phase cost?"""
    rects = [(128, src_x, T - 256, w, r * (T - 256), c * w) for r in range(G) for c in range(G)]
    wc, hc = G * w, G * (T - 256)
    plan = native.FusePlan(np.array(rects), T, T, hc, wc)
    canvas = torch.empty((P, hc, wc), dtype=torch.uint16, device=dev)
    ms = min(timeit(lambda: native.fuse_planes(plan, tiles, canvas)) for _ in range(2))
    alg = P * plan.covered_voxels * 4
    print(f'{name:58s} src_x {src_x:4d}: {ms:7.3f} ms {alg/ms/1e6:7.1f} GB/s', flush=True)
    del canvas


for sx in (128, 129, 130, 132, 136, 144, 160, 192, 122):
    run_src('aligned seams, source phase only', sx)

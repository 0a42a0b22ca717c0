"""Debug: the failing case of test_plane_groups_on_line_aligned_canvases (seed 0, feather, queues) with the mismatches located."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from image_stitcher_amd import native
from oracle import stitch_oracle as O
seed, mode, queues = int(sys.argv[1]) if len(sys.argv) > 1 else 0, native.SQ_FUSE_FEATHER, True
dev = torch.device('cuda:0')
rng = np.random.default_rng(4200 + seed)
th, tw = int(rng.integers(20, 70)), int(rng.integers(40, 260))
rows, cols = int(rng.integers(1, 4)), int(rng.integers(1, 4))
oy, ox = int(rng.integers(2, th // 3)), int(rng.integers(2, tw // 3))
n = rows * cols
crop = False
rects = np.zeros((n, 6), dtype=np.int64)
for r in range(rows):
    for c in range(cols):
        rects[r * cols + c] = (0, 0, th, tw, r * (th - oy) + c * 2, c * (tw - ox) + (rows - 1 - r) * 3)
ch = int(rects[:, 4].max() + th + rng.integers(0, 9))
cw = int(rects[:, 5].max() + tw + rng.integers(0, 9))
planes = int(rng.integers(1, 13))
tiles = rng.integers(0, 65536, size=(planes, n, th, tw)).astype(np.uint16)
gains = [np.exp(rng.normal(0, 0.4, size=(th, tw))).astype(np.float32) for _ in range(3)]
gains[2][rng.integers(0, th), rng.integers(0, tw)] = 0.0
gains[1][rng.integers(0, th), rng.integers(0, tw)] = 2.0 ** -30
which = [int(rng.integers(0, 3)) if rng.random() > 0.1 else -1 for _ in range(planes)]
if planes >= 7:
    which[:6] = [0] * 6
print(f'tile {th}x{tw}, grid {rows}x{cols}, overlap {oy},{ox}, canvas {ch}x{cw}, {planes} planes, gain image of each {which}')
print('rects', rects.tolist())
d_gains = [torch.from_numpy(g).to(dev) for g in gains]
flats = [None if k < 0 else d_gains[k] for k in which]
plan = native.FusePlan(rects, th, tw, ch, cw, mode)
d_tiles = torch.from_numpy(tiles).to(dev)
WANT = [O.fuse_plane_feather(list(tiles[p]), rects, ch, cw, None if which[p] < 0 else gains[which[p]], out_dtype=np.uint16) for p in range(planes)]
for flags, name in ((native.SQ_FUSE_FORCE_QUEUES, 'queues'), (native.SQ_FUSE_FORCE_STATIC, 'static')):
    for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
        aligned = native.empty_canvas(planes, ch, cw, torch.uint16, dev)
        aligned.view(torch.int16).fill_(-1)
        native.fuse_planes(plan, d_tiles, aligned, flats, flags=flags)
        torch.cuda.synchronize()
        cover = np.zeros((ch, cw), np.int32)
        for top, left, h, w, y, x in rects:
            cover[y:y + h, x:x + w] += 1
        for p in range(planes):
            want = WANT[p]
            got = aligned[p].cpu().numpy()
            bad = np.argwhere(got != want)
            if len(bad):
                print(f'{name} rep {rep} plane {p} (gain {which[p]}): {len(bad)} wrong:', [(int(y), int(x), int(cover[y, x]), int(got[y, x]), int(want[y, x])) for y, x in bad[:12]])
print('done')

"""The headline job's registration batch on its own (32 x 32 grid: 992 horizontal + 992 vertical pairs), a few times --
the program rocprofv3 is pointed at for per-kernel times and counters of the registration kernels."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, registration
dev = torch.device('cuda:0')
g, T = 32, 2048
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tiles = torch.randint(0, 65535, (256, T, T), dtype=torch.int32, device=dev).to(torch.uint16)
mm = native.tile_minmax(tiles)
(hp, (h0, h1)), (vp, (v0, v1)) = registration.all_pairs(g, g, T, T, 256, 256)
for batch in (hp, vp):
    batch['ref_tile'] %= len(tiles)
    batch['mov_tile'] %= len(tiles)
for _ in range(reps):
    a = native.register_pairs_async(tiles, mm, hp, h0, h1, 10, native.SQ_NORM_PHASE)
    b = native.register_pairs_async(tiles, mm, vp, v0, v1, 10, native.SQ_NORM_PHASE)
    ra, rb = a.fetch(), b.fetch()
torch.cuda.synchronize()
print('done', len(ra), len(rb), (h0, h1), (v0, v1))

"""Times the headline job's two registration batches (992 horizontal + 992 vertical pairs of a 32 x 32 grid) with events;
used with experiment builds (SQ_LIB_PATH + SQ_REG_* knobs) to compare launch shapes of the registration kernels."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, registration
dev = torch.device('cuda:0')
g, T = 32, 2048
tiles = torch.randint(0, 65535, (256, T, T), dtype=torch.int32, device=dev).to(torch.uint16)
mm = native.tile_minmax(tiles)
(hp, (h0, h1)), (vp, (v0, v1)) = registration.all_pairs(g, g, T, T, 256, 256)
for batch in (hp, vp):
    batch['ref_tile'] %= len(tiles)
    batch['mov_tile'] %= len(tiles)
out = []
for name, pairs, n0, n1 in (('h', hp, h0, h1), ('v', vp, v0, v1)):
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        p = native.register_pairs_async(tiles, mm, pairs, n0, n1, 10, native.SQ_NORM_PHASE)
        e1.record()
        r = p.fetch()
        best = min(best, e0.elapsed_time(e1))
    out.append(f'{name} {n0}x{n1}: {best:6.2f} ms')
print(' | '.join(out), '|', ' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('SQ_REG')))

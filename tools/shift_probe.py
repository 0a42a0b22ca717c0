"""How much does source/destination phase cost?  Giant single-tile copies with pixel shifts."""
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
H, W = 32768, 32768
tile = torch.empty((1, 1, H, W), dtype=torch.uint16, device=dev); tile.fill_(7)
os.environ['SQ_FUSE_FLAGS'] = os.environ.get('SQ_FUSE_FLAGS', '2')
for sx in (0, 1, 2, 4, 8, 16, 32, 64, 3):
    for dx in (0, 3):
        w = W - 64 - 8
        plan = native.FusePlan(np.array([(0, sx, H, w, 0, dx)]), H, W, H, W)
        canvas = torch.empty((1, H, W), dtype=torch.uint16, device=dev)
        ms = timeit(lambda: native.fuse_planes(plan, tile, canvas))
        print(f'src shift {sx:3d} px, dst shift {dx} px: {ms:.3f} ms -> {(H*w*4 + H*(W-w)*2)/ms/1e6:.1f} GB/s', flush=True)
        del canvas

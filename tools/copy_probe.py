"""Device copy ceiling: torch copy_ of large buffers and a giant single-tile fusion."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native

dev = torch.device('cuda:0')
n = 8 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev); a.fill_(3)
b = torch.empty(n, dtype=torch.uint8, device=dev)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    return min(x.elapsed_time(y) for x, y in evs)
ms = timeit(lambda: b.copy_(a))
print(f'torch copy_ 8 GiB: {ms:.3f} ms -> {2*n/ms/1e6:.1f} GB/s (read+write)')
ms = timeit(lambda: b.zero_())
print(f'torch zero_ 8 GiB: {ms:.3f} ms -> {n/ms/1e6:.1f} GB/s (write)')
del a, b
# one giant "tile" copied 1:1 (aligned) and with a 3-pixel shift (misaligned both ways)
H, W = 32768, 32768
tile = torch.empty((1, 1, H, W), dtype=torch.uint16, device=dev); tile.fill_(7)
for name, rect, ch, cw in (('aligned 1:1', (0, 0, H, W, 0, 0), H, W), ('shifted by 3 px', (0, 3, H, W - 3, 0, 0), H, W - 3),
                           ('odd pitch', (0, 0, H, W - 5, 0, 0), H, W - 5)):
    plan = native.FusePlan(np.array([rect]), H, W, ch, cw)
    canvas = torch.empty((1, ch, cw), dtype=torch.uint16, device=dev)
    for flags in (3, 0, 1, 2):
        os.environ['SQ_FUSE_FLAGS'] = str(flags)
        ms = timeit(lambda: native.fuse_planes(plan, tile, canvas))
        print(f'giant tile {name} flags={flags}: {ms:.3f} ms -> {ch*cw*4/ms/1e6:.1f} GB/s')
    del canvas

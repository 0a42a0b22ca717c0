"""Time sq_downsample2 on a config-3-sized canvas plane batch (HIP events) and report the HBM rate.
Algorithmic bytes: odd source rows read whole + destination written = (1/2 + 1/4) of the source bytes."""
import sys
import time

import torch

sys.path.insert(0, '.')
from image_stitcher_amd import native

n, h, w = 4, 36428, 29108
a = torch.empty((n, h, w), dtype=torch.uint16, device='cuda')
a.view(torch.int16).random_(-30000, 30000)
levels = []
src = a
for lv in range(1, 7):
    out = torch.empty((n, src.shape[1] // 2, src.shape[2] // 2), dtype=torch.uint16, device='cuda')
    for _ in range(2):
        native.downsample2(src, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        native.downsample2(src, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = src.numel() * 2 * 0.75
    print(f"level {lv}: {tuple(src.shape)} -> {tuple(out.shape)}  {ms:.3f} ms  {alg / ms / 1e6:.0f} GB/s algorithmic "
          f"({alg / ms / 1e6 / 8000:.3f} of 8 TB/s)")
    src = out

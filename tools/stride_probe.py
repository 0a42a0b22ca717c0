"""The REAL fusion kernel (fuse_overwrite_zg_kernel, config-3 geometry, float32 gains) against the byte distance between
consecutive PLANES of the tile stacks and of the canvas, everything inside ONE pair of allocations in ONE process
(tools/membw_gains showed the rate of the plane-group access pattern to be bimodal in exactly that distance).
    python tools/stride_probe.py [planes=10] [grid=16]
Prints one line per (tile-plane pad, canvas-plane pad): ms per launch and the fraction of 8 TB/s."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from image_stitcher_amd import native, placement, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T = 2048
dev = torch.device('cuda:0')
spec, truth, wc, hc, xs, ys, order, order_rc = bench.grid_setup(g, 3000)
rects = placement.grid_rects(g, g, T, T, truth, order=order_rc)
plan = native.FusePlan(rects, T, T, hc, wc, native.SQ_FUSE_OVERWRITE)
Z = P // 2 if P >= 10 else P
MAXPAD = 16 << 20
plane_in = g * g * T * T * 2
plane_out = -(-(hc * wc * 2) // 128) * 128
tiles_flat = torch.empty(P * (plane_in + MAXPAD), dtype=torch.uint8, device=dev)
canvas_flat = torch.empty(P * (plane_out + MAXPAD), dtype=torch.uint8, device=dev)
tiles_flat.view(torch.int16)[:] = 1234          # any pixels: the kernel's rate does not depend on the values
ffs = [torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32) * np.float32(1 + 0.03125 * c)).to(dev) for c in range(2)]
flat_list = [ffs[p // Z] for p in range(P)]
flat_ptrs = native.pointer_table(flat_list, dev)
alg = bench.algorithmic_bytes(P, plan.covered_voxels, hc, wc, True)
tile_order = torch.tensor(order, dtype=torch.int64)


def run(tpad, cpad, reps=4):
    tstride, cstride = plane_in + tpad, plane_out + cpad
    ptrs = (tiles_flat.data_ptr() + torch.arange(P, dtype=torch.int64)[:, None] * tstride + tile_order[None, :] * (T * T * 2)).reshape(-1).to(dev)
    canvas = canvas_flat.view(torch.uint16).as_strided((P, hc, wc), (cstride // 2, wc, 1))
    best = 1e9
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.fuse_planes(plan, None, canvas, flat_list, tile_ptrs=ptrs, flat_ptrs=flat_ptrs)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print(f'{P} planes ({Z} per gain image) of the {g}x{g} grid, canvas {hc}x{wc}; tile plane {plane_in} B, canvas plane {plane_out} B', flush=True)
pads_t = [0, 128, 256, 512, 1024, 4096, 4352, 65792, 1782016, 3635456]
pads_c = [0, 128, 256, 512, 1024, 1536, 4096, 65792, 1289984, 2890496]
for tp in pads_t:
    row = []
    for cp in pads_c:
        ms = run(tp, cp)
        row.append(alg / ms / 1e6 / 8000)
    print(f'tile pad {tp:>8}: ' + ' '.join(f'{v:.3f}' for v in row) + f'   (canvas pads {pads_c})', flush=True)

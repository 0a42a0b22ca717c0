"""Registration of the bench's centre pairs for the seeds ranks 0..7 would use, plus a
PCIe-inclusive timing of one plane (pinned host tiles in, host canvas out)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, placement, registration, synth
dev = torch.device('cuda:0')
g, T, OV = 16, 2048, 244
for rank in range(8):
    spec = synth.GridSpec(rows=g, cols=g, tile_h=T, tile_w=T, ov_y=OV, ov_x=OV, jy=3, jx=-2,
                          channels=synth.DEFAULT_CHANNELS[:4], nz=10, seed=1000 * 3 + rank * 100)
    desc = np.zeros(g * g, dtype=native.SYNTH_DTYPE)
    for r in range(g):
        for c in range(g):
            oy, ox = spec.origin(r, c)
            desc[r * g + c] = (spec.scene_seed(0, 0, 0, 0) % 2**64, spec.noise_seed(0, 0, 0, 0, spec.fov_index(r, c)) % 2**64, oy, ox)
    tiles = native.synth_tiles(desc, T, T, spec.noise, 'uint16', dev)
    xs = [spec.stage_mm(0, c)[0] for c in range(g)]
    ys = [spec.stage_mm(r, 0)[1] for r in range(g)]
    s = registration.register_grid_center(tiles, g, g, xs, ys, spec.pixel_size_um, spec.pixel_binning, 'phase')
    print(f'rank-seed {rank}: h={s.h_shift} v={s.v_shift}', flush=True)
    assert s.h_shift == (3, -OV) and s.v_shift == (-OV, -2)
# PCIe-inclusive: one plane, pinned host buffers
shifts = placement.Shifts((3, -OV), (-OV, -2))
rects = placement.grid_rects(g, g, T, T, shifts)
wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=shifts)
plan = native.FusePlan(rects, T, T, hc, wc)
host_in = torch.empty((1, g * g, T, T), dtype=torch.uint16).pin_memory()
host_in.copy_(tiles.cpu()[None])
host_out = torch.empty((1, hc, wc), dtype=torch.uint16).pin_memory()
canvas = torch.empty((1, hc, wc), dtype=torch.uint16, device=dev)
dtiles = torch.empty_like(host_in, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dtiles.copy_(host_in, non_blocking=True)
    native.fuse_planes(plan, dtiles, canvas)
    host_out.copy_(canvas, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'PCIe-inclusive plane: {dt*1e3:.1f} ms -> {hc*wc/dt/1e6:.0f} Mvoxel/s '
          f'({(host_in.numel()*2 + host_out.numel()*2)/dt/1e9:.1f} GB/s over the link)', flush=True)

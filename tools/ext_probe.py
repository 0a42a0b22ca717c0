"""Throughput of the extensions: feather fusion and all-pairs registration on a 16x16 grid."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, placement, registration, synth
dev = torch.device('cuda:0')
g, T, OV, P = 16, 2048, 244, int(os.environ.get('SQ_EXT_PLANES', '4'))
spec = synth.GridSpec(rows=g, cols=g, tile_h=T, tile_w=T, ov_y=OV, ov_x=OV, seed=5)
tiles = torch.empty((P, g * g, T, T), dtype=torch.uint16, device=dev)
for p in range(P):
    desc = np.zeros(g * g, dtype=native.SYNTH_DTYPE)
    for r in range(g):
        for c in range(g):
            oy, ox = spec.origin(r, c)
            desc[r * g + c] = (spec.scene_seed(0, 0, p, 0) % 2**64, spec.noise_seed(0, 0, p, 0, r * g + c) % 2**64, oy, ox)
    native.synth_tiles(desc, T, T, 200, 'uint16', dev, out=tiles[p])
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
shifts = placement.Shifts((3, -OV), (-OV, -2))
wc, hc = placement.canvas_size(g, g, T, T, use_registration=True, shifts=shifts)
rects = placement.grid_rects(g, g, T, T, shifts, crop=False)
plan = native.FusePlan(rects, T, T, hc, wc, native.SQ_FUSE_FEATHER)
print(f'feather plan: {plan.n_spans} spans, {plan.n_items} items, max refs {plan.max_refs}')
ff = torch.from_numpy(synth.synthetic_flatfield(T, T, np.float32)).to(dev)
rho = g * g * T * T / (hc * wc)
for dt, flats in ((torch.uint16, None), (torch.uint16, [ff] * P), (torch.float32, None)):
    canvas = native.empty_canvas(P, hc, wc, dt, dev)
    t = timeit(lambda: native.fuse_planes(plan, tiles, canvas, flats, flags=int(os.environ.get('SQ_EXT_FLAGS', '0'))))
    alg = P * hc * wc * (2 * rho + canvas.element_size())       # 2*rho B read + the voxel written
    print(f'feather -> {dt}{" with float32 gains" if flats else ""}: {t*1e3:.2f} ms for {P} planes -> '
          f'{P*hc*wc/t/1e6:.0f} Mvoxel/s, {alg/t/1e9:.0f} GB/s algorithmic ({alg/t/8e12:.3f} of 8 TB/s)')
    if flats and os.environ.get('SQ_EXT_AB'):   # plane groups against one plane at a time, same buffers, alternating
        for rnd in range(3):
            for fl in (0, native.SQ_FUSE_NO_PLANE_GROUPS):
                t = timeit(lambda: native.fuse_planes(plan, tiles, canvas, flats, flags=fl))
                print(f'  round {rnd} flags {fl}: {t*1e3:.2f} ms ({alg/t/8e12:.3f} of 8 TB/s)', flush=True)
    del canvas
if os.environ.get('SQ_EXT_FEATHER_ONLY'):
    sys.exit(0)
xs = [spec.stage_mm(0, c)[0] for c in range(g)]
ys = [spec.stage_mm(r, 0)[1] for r in range(g)]
mx, my = placement.registration_crop_widths(xs, ys, T, T, spec.pixel_size_um, spec.pixel_binning)
(hp, (h0, h1)), (vp, (v0, v1)) = registration.all_pairs(g, g, T, T, mx, my)
mm = native.tile_minmax(tiles[0])
t = timeit(lambda: native.tile_minmax(tiles[0]))
print(f'tile_minmax of {g*g} tiles: {t*1e3:.2f} ms -> {g*g*T*T*2/t/1e9:.0f} GB/s')
for name, pairs, n0, n1 in (('horizontal', hp, h0, h1), ('vertical', vp, v0, v1)):
    t = timeit(lambda: registration.register_pairs(tiles[0], pairs, n0, n1, 10, 'phase', mm))
    s, e, _ = registration.register_pairs(tiles[0], pairs, n0, n1, 10, 'phase', mm)
    conv = registration.horizontal_shift_from if name == 'horizontal' else registration.vertical_shift_from
    got = {conv(x, n1 if name == 'horizontal' else n0) for x in s}
    print(f'{name}: {len(pairs)} pairs of {n0}x{n1}: {t*1e3:.2f} ms -> {len(pairs)/t:.0f} pairs/s; distinct shifts {got}')

"""Rates of the secondary kernels on the GPU box (not the bench contract): the chunk encoder (Blosc-1 / LZ4) on smooth,
noisy and zero planes; the BaSiC fit; registration batches with power-of-two and Bluestein crop sides."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_stitcher_amd import native, placement, registration, synth

dev = torch.device('cuda:0')


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


only_reg = len(sys.argv) > 1 and sys.argv[1] == 'registration'
# ---- chunk encoder ------------------------------------------------------------------------------------------------
H, W, P = 8192, 8192, 8
rng = np.random.default_rng(0)
smooth = (2000 + 3000 * synth.synthetic_flatfield(H, W, np.float32))
kinds = {
    'smooth + 8 counts of noise (microscope-like)': lambda: (smooth + rng.normal(0, 8, (H, W))).astype(np.uint16),
    'hash noise (the bench tiles: incompressible)': lambda: synth.scene_patch(3, 0, 0, H, W).astype(np.uint16),
    'zeros': lambda: np.zeros((H, W), np.uint16),
}
for name, make in ({} if only_reg else kinds).items():
    plane = torch.from_numpy(make()).to(dev)
    planes = plane[None].repeat(P, 1, 1).contiguous()
    buf = native.BloscBuffers(P, H, W, np.uint16, 512, 512, dev)
    ms = timed(lambda: native.blosc_encode_planes(planes, 512, 512, buf))
    total = int(buf.offsets[-1].item())
    raw = planes.numel() * 2
    print(f'blosc encode, {name}: {ms:.2f} ms for {raw / 1e9:.2f} GB -> {raw / ms / 1e6:.0f} GB/s in, ratio {raw / max(total, 1):.2f}', flush=True)
    del planes, buf

# ---- BaSiC ---------------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_basic_oracle_cpu import planted_stack
for n, h, w in (() if only_reg else ((48, 2048, 2048), (33, 512, 512))):
    stack, gain = planted_stack(n, h, w, seed=1, objects=max(6, 30 * h * w // (256 * 320)))
    t = torch.from_numpy(stack).to(dev)
    native.basic_fit(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    flat, info = native.basic_fit(t)
    dt = time.perf_counter() - t0
    err = np.abs(flat.cpu().numpy() / gain - 1)
    print(f'basic_fit {n} x {h}x{w}: {dt * 1e3:.1f} ms, {info}, planted-gain error mean {err.mean():.4f}', flush=True)

# ---- registration batches -----------------------------------------------------------------------------------------
for (th, tw, ov, label) in ((2048, 2048, 244, 'power-of-two crops 1024 x 256'), (4168, 6244, 300, 'Bluestein crops (6244 x 4168 sensor)'),
                            (3000, 3000, 288, 'mixed-radix crops (3000 x 3000 sensor: 1500 = 2^2 3 5^3)'),
                            (6380, 9568, 300, 'long Bluestein crops (9568 x 6380 sensor: 4784, 3190)'),
                            (14192, 10640, 300, 'lines in the workspace (a 14192 x 10640 sensor: 7096 -> a Bluestein line of 14 k points, 5320 = 2^3 5 7 19)')):
    g = 6 if th * tw <= 30e6 else (4 if th * tw <= 70e6 else 3)
    tiles = torch.from_numpy(rng.integers(0, 65535, (g * g, th, tw)).astype(np.uint16)).to(dev)
    mm = native.tile_minmax(tiles)
    (hp, (h0, h1)), (vp, (v0, v1)) = registration.all_pairs(g, g, th, tw, ov + 12, ov + 12)
    for pairs, n0, n1 in ((hp, h0, h1), (vp, v0, v1)):
        code = native.SQ_NORM_PHASE
        ms = timed(lambda: native.register_pairs_async(tiles, mm, pairs, n0, n1, 10, code))
        print(f'register {len(pairs)} pairs, {label}, crop {n0} x {n1}: {ms:.2f} ms -> {len(pairs) / ms * 1e3:.0f} pairs/s', flush=True)
    del tiles

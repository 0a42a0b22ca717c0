"""Registration of the headline job's batch (992 pairs of 1024 x 256) in ONE launch chain against chunks of k pairs
(one chain per chunk, enqueued back to back): do the spectra (4.2 MB per pair) survive in the 256 MB Infinity Cache
between the row, column and inverse kernels when the chunk is small enough?"""
import os, sys, time
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


def run(pairs, n0, n1, chunk):
    pend = [native.register_pairs_async(tiles, mm, pairs[i:i + chunk], n0, n1, 10, native.SQ_NORM_PHASE) for i in range(0, len(pairs), chunk)]
    return np.concatenate([p.fetch() for p in pend])


for name, pairs, n0, n1 in (('horizontal 1024 x 256', hp, h0, h1), ('vertical 256 x 1024', vp, v0, v1)):
    want = run(pairs, n0, n1, len(pairs))
    for chunk in (len(pairs), 496, 248, 124, 62, 48, 32, 16):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            got = run(pairs, n0, n1, chunk)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        assert np.array_equal(got, want), 'chunked results differ'
        print(f'{name}: {len(pairs)} pairs in chunks of {chunk:4d}: {best * 1e3:7.2f} ms -> {len(pairs) / best:7.0f} pairs/s  (spectra of a chunk: {chunk * 2 * n0 * (n1 // 2 + 1) * 16 / 2**20:6.0f} MiB)', flush=True)

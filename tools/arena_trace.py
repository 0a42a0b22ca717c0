"""What the phases of an arena's creation cost (SQ_ARENA_TRACE=1 python tools/arena_trace.py [GiB=80] [slice MiB=64 ...])."""
import sys, time, torch
sys.path.insert(0, '.')
from image_stitcher_amd import native
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 80
for mib in ([int(x) for x in sys.argv[2:]] or [64]):
    t0 = time.time()
    a = native.DeviceArena(gib << 30, torch.device('cuda:0'), slice_bytes=mib << 20)
    t1 = time.time()
    print(f'slices of {mib} MiB: created in {t1 - t0:.2f} s', a.info, flush=True)
    a.close()
    print(f'closed in {time.time() - t1:.2f} s', flush=True)

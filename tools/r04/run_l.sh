#!/bin/bash
# r04 l: the arena with incremental candidates and re-derived classes: tests, then arenas of several sizes with the trace
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_arena_gpu.py -x -q > $O/test_arena2.log 2>&1 || { echo tests failed; tail -40 $O/test_arena2.log; exit 1; }
tail -2 $O/test_arena2.log
SQ_ARENA_TRACE=1 timeout -k 10 600 python3 - > $O/arena_sizes.log 2>&1 <<'PY' || { echo failed; tail -20 $O/arena_sizes.log; exit 1; }
import sys, torch
sys.path.insert(0, '.')
from image_stitcher_amd import native
dev = torch.device('cuda:0')
for gib in (8, 40, 80, 80, 160):
    a = native.DeviceArena(gib << 30, dev)
    print(gib, 'GiB ->', a.info, flush=True)
    a.close()
PY
grep -v "amdgpu.ids\|with the reference" $O/arena_sizes.log | cut -c1-900

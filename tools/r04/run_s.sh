#!/bin/bash
# r04 s: the arena with one hipMemUnmap per mapping: the plain-C caller, the arena tests (with the trace: what the per-slice calls cost)
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_c_abi_gpu.py tests/test_arena_gpu.py -x -q 2>&1 | grep -E "differs|passed|failed|Error" | head -12
SQ_ARENA_TRACE=1 timeout -k 10 300 python3 -c "
import sys, torch
sys.path.insert(0, '.')
from image_stitcher_amd import native
a = native.DeviceArena(80 << 30, torch.device('cuda:0')); print(a.info); a.close()
a = native.DeviceArena(80 << 30, torch.device('cuda:0')); print(a.info['create_ms'], a.info['class_slices']); a.close()
" 2>&1 | grep -v "amdgpu.ids\|units in creation" | cut -c1-260

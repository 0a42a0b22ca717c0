#!/bin/bash
# exp24: uint8 tiles through the per-plane kernels (no plane groups / seam owners for uint8): where do they stand?
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
echo "== uint8, no gains, 4x4 check then 16 planes"; timeout -k 10 300 python tools/fuse_probe.py --u8 --grid 4 --planes 3 --steps 2 --check
timeout -k 10 300 python tools/fuse_probe.py --u8 --planes 16 --steps 5
echo "== uint8, float32 gains, 16 planes of 2 channels"; timeout -k 10 300 python tools/fuse_probe.py --u8 --planes 16 --nflats 2 --flat f32 --steps 5
} > gpurun_out/r2_exp24.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp24.log

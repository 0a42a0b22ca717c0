#!/bin/bash
# exp17: groups of ONE plane through the plane-group code (seam owners, unconditional loads) against the per-plane pipeline
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
echo "== 4 planes, 4 gain images (groups of one)"; timeout -k 10 300 python tools/fuse_probe.py --planes 4 --nflats 4 --flat f32 --steps 5 --libs default,singles --check
echo "== 8 planes, 4 gain images (groups of two: both builds the same code)"; timeout -k 10 300 python tools/fuse_probe.py --planes 8 --nflats 4 --flat f32 --steps 5 --libs default,singles
} > gpurun_out/r2_exp17.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp17.log

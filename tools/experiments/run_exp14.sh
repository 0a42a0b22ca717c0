#!/bin/bash
# exp14: item height and queue chunk with plane groups + seam owners, variants alternated in one process on the same buffers
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
echo "== cfg3 grid, 20 planes, 2 channels"
timeout -k 10 400 python tools/fuse_probe.py --grid 16 --planes 20 --nflats 2 --flat f32 --steps 5 --libs default,rows4,rows16,chunk4,chunk16 --check
echo "== cfg4 grid, 10 planes"
timeout -k 10 400 python tools/fuse_probe.py --grid 32 --planes 10 --flat f32 --steps 4 --libs default,rows4,rows16,chunk4,chunk16
} > gpurun_out/r2_exp14.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp14.log

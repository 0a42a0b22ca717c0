#!/bin/bash
# exp26b: zero-fill items dealt into the XCD lanes (running beside the copies) instead of after them
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
SQ_PLAN_ORDER=5 timeout -k 10 400 python tools/fuse_probe.py --grid 4 --planes 6 --flat f32 --steps 2 --libs expo --check
SQ_PLAN_ORDER=5 timeout -k 10 400 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --steps 5 --libs default,expo
} > gpurun_out/r2_exp26b.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp26b.log

#!/bin/bash
# exp26: zero fill with plain instead of non-temporal stores
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --steps 5 --libs default,zp > gpurun_out/r2_exp26.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp26.log

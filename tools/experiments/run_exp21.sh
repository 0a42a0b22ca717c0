#!/bin/bash
# exp21: cache policy of the plane-group kernel's 16-byte stores: nt (shipped) / plain / sc1 nt / sc0 sc1
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
timeout -k 10 400 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --steps 4 --libs default,st1,st2,st3
} > gpurun_out/r2_exp21.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp21.log

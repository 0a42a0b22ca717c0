#!/bin/bash
# exp22: what seam owners are worth for small groups (2 planes per gain image) and for the no-gain per-plane kernel's bound
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
echo "== 8 planes, 4 gain images (groups of two): seam owners (flags 0) against both-write (flags 8)"
timeout -k 10 300 python tools/fuse_probe.py --planes 8 --nflats 4 --flat f32 --steps 5 --ab 8
echo "== 16 planes, no gains (per-plane pipeline, no seam owners there): reference rate"
timeout -k 10 300 python tools/fuse_probe.py --planes 16 --steps 5
} > gpurun_out/r2_exp22.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp22.log

#!/bin/bash
# exp20: ONE row per workgroup (wave w = slot w), one-shot launch, groups of 5 / 3 / 2 planes -- the "few memory operations
# per thread" regime of r02_membw_2d.log on the real kernel -- against the shipped persistent kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
{
echo "== check: 4x4 grid, 7 planes, row-per-workgroup builds, one-shot"
timeout -k 10 300 python tools/fuse_probe.py --grid 4 --planes 7 --flat f32 --steps 2 --flags 2 --blocks 2147483647 --libs rw5,rw2 --check
echo "== persistent (shipped) kernel: groups of 5 (default) and of 2"
timeout -k 10 300 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --steps 4 --libs default,zb2
echo "== one-shot, one row per workgroup: groups of 5 / 3 / 2; the shipped build one-shot beside them"
timeout -k 10 400 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --steps 4 --flags 2 --blocks 2147483647 --libs default,rw5,rw3,rw2
echo "== the same builds through the persistent grid with device queues (flags 1)"
timeout -k 10 400 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --steps 4 --flags 1 --libs default,rw5,rw2
} > gpurun_out/r2_exp20.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp20.log

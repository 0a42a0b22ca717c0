#!/bin/bash
# exp16: feather mode with plane groups (blend strips and one-tile items through the planes of a channel together);
# waves per SIMD of the grouped kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fuse_gpu.py -x -q -m gpu > gpurun_out/r2_exp16_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r2_exp16_tests.log; [ $rc = 0 ] || exit 1
{
echo "== feather, 4x4 grid, 7 planes, gains: check"; timeout -k 10 300 python tools/fuse_probe.py --feather --grid 4 --planes 7 --flat f32 --steps 3 --check
echo "== feather, 4x4 grid, 7 planes, no gains: check"; timeout -k 10 300 python tools/fuse_probe.py --feather --grid 4 --planes 7 --steps 3 --check
echo "== feather, gains, 10 planes: default (flags 0) against one plane at a time (flags 4)"; timeout -k 10 300 python tools/fuse_probe.py --feather --planes 10 --flat f32 --steps 4 --ab 4
echo "== feather, no gains, 10 planes: default against one plane at a time"; timeout -k 10 300 python tools/fuse_probe.py --feather --planes 10 --steps 4 --ab 4
echo "== feather, gains, 10 planes: waves per SIMD asked of the compiler 1 (default) / 3 / 4"; timeout -k 10 300 python tools/fuse_probe.py --feather --planes 10 --flat f32 --steps 4 --libs default,fz3,fz4
echo "== feather, no gains, 10 planes: the same"; timeout -k 10 300 python tools/fuse_probe.py --feather --planes 10 --steps 4 --libs default,fz3,fz4
echo "== overwrite, gains, 20 planes: plane-group kernel at 4 (default, 97 VGPRs) / 5 waves"; timeout -k 10 300 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --steps 5 --libs default,oz5
} > gpurun_out/r2_exp16_feather.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp16_feather.log

#!/bin/bash
# exp18: plane groups for float64 gains (overwrite mode): parity tests, then default against one plane at a time
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fuse_gpu.py -x -q -m gpu > gpurun_out/r2_exp18_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r2_exp18_tests.log; [ $rc = 0 ] || exit 1
{
echo "== float64 gains, 4x4 grid, 7 planes: check"; timeout -k 10 300 python tools/fuse_probe.py --grid 4 --planes 7 --flat f64 --steps 3 --check
echo "== float64 gains, 16x16 grid, 10 planes: default (flags 0) against one plane at a time (flags 4)"; timeout -k 10 300 python tools/fuse_probe.py --planes 10 --flat f64 --steps 4 --ab 4
echo "== float64 gains, 20 planes, 2 channels"; timeout -k 10 300 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f64 --steps 4 --ab 4
} > gpurun_out/r2_exp18.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp18.log

#!/bin/bash
# exp13: seam owners (one writer per 128-byte line at vertical seams) against both-write, same process and buffers
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_fuse_gpu.py -x -q -m gpu > gpurun_out/r2_exp13_tests.log 2>&1 || { tail -30 gpurun_out/r2_exp13_tests.log; exit 1; }
tail -3 gpurun_out/r2_exp13_tests.log
{
echo "== cfg3 grid, 10 planes of one channel (flags 8 = SQ_FUSE_NO_SEAM_OWNERS)"
timeout -k 10 300 python tools/fuse_probe.py --grid 16 --planes 10 --flat f32 --steps 6 --ab 8 --check
echo "== cfg3 grid, 40 planes, 4 channels"
timeout -k 10 300 python tools/fuse_probe.py --grid 16 --planes 40 --nflats 4 --flat f32 --steps 4 --ab 8
echo "== cfg4 grid, 10 planes"
timeout -k 10 300 python tools/fuse_probe.py --grid 32 --planes 10 --flat f32 --steps 4 --ab 8
} > gpurun_out/r2_exp13_ab.log 2>&1
cat gpurun_out/r2_exp13_ab.log

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_fuse_gpu.py -m gpu -x -q > $O/feather_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -12 $O/feather_tests.log; [ $rc = 0 ] || exit 1
{
echo "# tools/ext_probe.py, 16x16 grid of 2048^2 tiles, 10 planes, feather mode: plane groups (flags 0) against the per-plane kernel (flags 4 = SQ_FUSE_NO_PLANE_GROUPS), same process and buffers"
for fl in 0 4 0 4; do SQ_EXT_PLANES=10 SQ_EXT_FEATHER_ONLY=1 SQ_EXT_FLAGS=$fl timeout -k 10 300 python tools/ext_probe.py 2>&1 | grep "feather ->" | sed "s/^/flags $fl: /"; done
} > $O/exp_feather.log 2>&1
cat $O/exp_feather.log | cut -c1-220

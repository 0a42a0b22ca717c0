#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_fuse_gpu.py tests/test_stitcher_gpu.py -x -q -m gpu > $O/v_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/v_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 500 python tools/order_probe.py 16 4 10 2 nogain > $O/exp_group_placement_nogain.log 2>&1; echo "order probe nogain rc $?"; cat $O/exp_group_placement_nogain.log

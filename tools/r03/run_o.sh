#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 500 python tools/order_probe.py 16 4 10 3 > $O/exp_group_placement_cfg3.log 2>&1; echo "order probe cfg3 rc $?"; cat $O/exp_group_placement_cfg3.log
timeout -k 10 500 python tools/order_probe.py 32 1 10 3 > $O/exp_group_placement_cfg4_batch.log 2>&1; echo "order probe cfg4 batch rc $?"; cat $O/exp_group_placement_cfg4_batch.log
timeout -k 10 900 python -m pytest tests/test_fuse_gpu.py -x -q -m gpu > $O/o_tests_fuse.log 2>&1; rc=$?; echo "pytest fuse rc $rc"; tail -5 $O/o_tests_fuse.log

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_plan_gpu.py -m gpu -x -q > $O/plan2_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 $O/plan2_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python tools/plan_time.py > $O/plan2_time.log 2>&1; echo "rc $?"; tail -12 $O/plan2_time.log

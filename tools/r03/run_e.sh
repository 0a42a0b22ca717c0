#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 600 python tools/stride_probe.py 10 16 > $O/exp_stride_probe_cfg3.log 2>&1; echo "stride probe rc $?"; cat $O/exp_stride_probe_cfg3.log
timeout -k 10 600 python -m pytest tests/test_register_gpu.py -x -q -m gpu > $O/e_tests_register.log 2>&1; rc=$?; echo "pytest register rc $rc"; tail -15 $O/e_tests_register.log

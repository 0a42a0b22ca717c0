#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_register_gpu.py tests/test_stitcher_gpu.py -x -q -m gpu > $O/swz_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/swz_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python tools/kernel_probe.py registration > $O/kernel_probe_registration_swz.log 2>&1; echo "probe rc $?"; cat $O/kernel_probe_registration_swz.log
SQ_BENCH_BREAKDOWN=1 timeout -k 10 300 python3 bench.py --workload cfg4 --planes 10 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "bench\]" | cut -c1-300

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_register_gpu.py -x -q -m gpu > $O/t_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/t_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python tools/kernel_probe.py registration > $O/kernel_probe_registration_v2.log 2>&1; echo "probe rc $?"; cat $O/kernel_probe_registration_v2.log
# the RCCL branch with ONE rank (communicator over the device, float64 pair-table all-gather, all-reduce, barriers)
SQ_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --workload cfg4 --planes 10 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_rccl_one_rank.json 2> $O/bench_cfg4_rccl_one_rank.err; echo "rccl one rank rc $?"; cut -c1-900 $O/bench_cfg4_rccl_one_rank.json; grep -i -E "nccl|rccl|error" $O/bench_cfg4_rccl_one_rank.err | head -5

#!/bin/bash
# full GPU suite, smoke, bench
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests_final.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -6 gpurun_out/r2_tests_final.log; [ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc $?"; cat gpurun_out/r2_bench_final.json

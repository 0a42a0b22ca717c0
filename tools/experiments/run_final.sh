#!/bin/bash
# round-2 final: full GPU suite, smoke, bench, rocprofv3 kernel stats of the bench
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests_final.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r2_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc $?"; cat gpurun_out/r2_bench_final.json
rm -rf gpurun_out/r2_prof_final
SQ_BENCH_NO_REFERENCE_JOB=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof_final -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_final_under_rocprof.json 2> gpurun_out/r2_prof_final.err || { echo rocprof failed; tail -5 gpurun_out/r2_prof_final.err; exit 1; }
f=$(find gpurun_out/r2_prof_final -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r2_kernel_stats_final.csv; head -6 gpurun_out/r2_kernel_stats_final.csv; cat gpurun_out/r2_bench_final_under_rocprof.json

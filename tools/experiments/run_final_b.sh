#!/bin/bash
# rocprofv3 kernel stats of the bench + PMC traffic passes (separate runs, --kernel-trace only beside --pmc)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
S="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
export SQ_BENCH_NO_REFERENCE_JOB=1
rm -rf gpurun_out/r2_prof_final gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof_final -- $B > gpurun_out/r2_bench_final_under_rocprof.json 2> gpurun_out/r2_prof_final.err || { echo rocprof failed; tail -5 gpurun_out/r2_prof_final.err; exit 1; }
f=$(find gpurun_out/r2_prof_final -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r2_kernel_stats_final.csv; head -6 gpurun_out/r2_kernel_stats_final.csv; cat gpurun_out/r2_bench_final_under_rocprof.json
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_pmc_fetch -- $S > gpurun_out/r2_pmc_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 gpurun_out/r2_pmc_fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_pmc_write -- $S > gpurun_out/r2_pmc_write.log 2>&1 || { echo pmc write failed; tail -5 gpurun_out/r2_pmc_write.log; exit 1; }
python tools/pmc_traffic.py gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write cfg3 40 "plane-group kernel with seam owners (round 2), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py --steps 2 --warmup 1" && cp profiles/pmc_traffic_latest.json gpurun_out/r2_pmc_traffic_latest.json

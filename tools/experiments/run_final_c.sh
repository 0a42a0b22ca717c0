#!/bin/bash
# the headline job's launch (10-plane batch of the 32x32 grid): rocprofv3 kernel stats + PMC traffic passes
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python3 bench.py --workload cfg4 --planes 40 --steps 2 --warmup 1 --no-cpu-baseline"
S="python3 bench.py --workload cfg4 --planes 20 --steps 1 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/r2_prof_cfg4 gpurun_out/r2_pmc4_fetch gpurun_out/r2_pmc4_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof_cfg4 -- $B > gpurun_out/r2_bench_cfg4_under_rocprof.json 2> gpurun_out/r2_prof_cfg4.err || { echo rocprof failed; tail -5 gpurun_out/r2_prof_cfg4.err; exit 1; }
f=$(find gpurun_out/r2_prof_cfg4 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r2_kernel_stats_cfg4.csv; head -4 gpurun_out/r2_kernel_stats_cfg4.csv; cat gpurun_out/r2_bench_cfg4_under_rocprof.json
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_pmc4_fetch -- $S > gpurun_out/r2_pmc4_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 gpurun_out/r2_pmc4_fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_pmc4_write -- $S > gpurun_out/r2_pmc4_write.log 2>&1 || { echo pmc write failed; tail -5 gpurun_out/r2_pmc4_write.log; exit 1; }
python tools/pmc_traffic.py gpurun_out/r2_pmc4_fetch gpurun_out/r2_pmc4_write cfg4 10 "one resident batch of the headline job: 10 planes of the 32x32 grid (plane groups, seam owners), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py --workload cfg4 --planes 20 --steps 1 --warmup 1" gpurun_out/r2_pmc_traffic_cfg4_batch.json

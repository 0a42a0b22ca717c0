#!/bin/bash
# kernel stats of the headline job's pieces (all-pairs registration of 1 984 pairs + one 10-plane batch)
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
rm -rf $O/prof_cfg4
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python3 bench.py --workload cfg4 --planes 10 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic > $O/bench_cfg4_under_rocprof.json 2> $O/prof_cfg4.err || { echo rocprof failed; tail -5 $O/prof_cfg4.err; exit 1; }
f=$(find $O/prof_cfg4 -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_cfg4.csv; head -16 $O/kernel_stats_cfg4.csv | cut -c1-200; cat $O/bench_cfg4_under_rocprof.json | cut -c1-1500
rm -rf $O/prof_cfg4
SQ_BENCH_BREAKDOWN=1 timeout -k 10 500 python3 bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_200planes.json 2> $O/bench_cfg4_200planes.err; echo "job rc $?"; cat $O/bench_cfg4_200planes.json | cut -c1-2500; tail -2 $O/bench_cfg4_200planes.err

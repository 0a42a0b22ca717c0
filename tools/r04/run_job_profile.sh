#!/bin/bash
# r04 job profile: rocprofv3 kernel stats of one resident batch of the headline job (10 planes of the 32 x 32 grid) on the last build
O=gpurun_out/r4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf $O/job_trace
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/job_trace -o run -- python3 bench.py --workload cfg4 --planes 10 --steps 3 --warmup 1 --no-cpu-baseline > $O/job_under_rocprof.json 2> $O/job_under_rocprof.err || { echo failed; tail -5 $O/job_under_rocprof.err; exit 1; }
cp $O/job_trace/run_kernel_stats.csv $O/bench_cfg4_10planes_kernel_stats.csv
grep "^{" $O/job_under_rocprof.json | tail -1 > $O/bench_cfg4_10planes_under_rocprof.json
head -12 $O/bench_cfg4_10planes_kernel_stats.csv | cut -c1-220
python3 -c "
import json; d=json.load(open('$O/bench_cfg4_10planes_under_rocprof.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d.get('host_ms_per_job_rank0'))"
rm -rf $O/job_trace

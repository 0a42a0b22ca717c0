#!/bin/bash
# r04 profile: rocprofv3 kernel stats of the bench on HEAD (kernel trace + stats only), then the SQ / TCC view of the fusion launch
# (counters only, one pass per set, the program directly after --).  FETCH_SIZE / WRITE_SIZE come from the bench's own child passes.
O=gpurun_out/r4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export SQ_BENCH_NO_REFERENCE_JOB=1
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-feather"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o run -- python3 $ARGS > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || { echo trace failed; tail -5 $O/bench_under_rocprof.err; exit 1; }
grep "^{" $O/bench_under_rocprof.json | tail -1 > $O/bench_under_rocprof_line.json
head -4 $O/bench_trace/run_kernel_stats.csv | cut -c1-220
pass() { local name=$1; shift
  timeout -k 10 600 rocprofv3 --pmc "$@" --output-format csv -d $O/bench_pmc_$name -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --no-feather > /dev/null 2> $O/bench_pmc_$name.err || { echo "pmc $name failed"; tail -5 $O/bench_pmc_$name.err; return 1; }
  echo "== pass $name: $*" >> $O/bench_counters.log
  python3 tools/r04/pmc_by_kernel.py $O/bench_pmc_$name fuse_overwrite >> $O/bench_counters.log; }
rm -f $O/bench_counters.log
pass sq SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES && pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum && pass mem FETCH_SIZE && pass mem2 WRITE_SIZE
cat $O/bench_counters.log
find $O -name "*kernel_trace.csv" -size +5M -delete

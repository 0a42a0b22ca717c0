#!/bin/bash
# r04 k: rocprofv3 evidence for the feather kernels (config-3 geometry, 40 planes uint16 / 20 planes float32, canvas in a mixed
# arena): kernel trace + stats, then counters only -- FETCH_SIZE, WRITE_SIZE, and the SQ view -- one pass each, program after --
O=gpurun_out/r4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/feather_trace -o run -- python3 tools/feather_probe.py 4 10 5 > $O/feather_trace.log 2>&1 || { echo trace failed; tail -5 $O/feather_trace.log; exit 1; }
grep -v amdgpu.ids $O/feather_trace.log | tail -9
pass() { local name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $O/feather_pmc_$name -o run -- python3 tools/feather_probe.py 4 10 2 > $O/feather_pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $O/feather_pmc_$name.log; return 1; }
  echo "== pass $name: $*" >> $O/feather_counters.log
  python3 tools/r04/pmc_by_kernel.py $O/feather_pmc_$name fuse_feather >> $O/feather_counters.log; }
rm -f $O/feather_counters.log
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass sq1 SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES && pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
cat $O/feather_counters.log
find $O -name "*_kernel_stats.csv" | head -3
find $O -name "*kernel_trace.csv" -size +5M -delete
# the headline job's resident batch (10 planes of the 32 x 32 grid): FETCH_SIZE / WRITE_SIZE passes for profiles/pmc_traffic_cfg4_batch.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/cfg4_pmc_$c -o run -- python3 bench.py --workload cfg4 --planes 20 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/cfg4_pmc_$c.err || { echo "cfg4 pmc $c failed"; tail -5 $O/cfg4_pmc_$c.err; exit 1; }
done
python3 tools/pmc_traffic.py $O/cfg4_pmc_FETCH_SIZE $O/cfg4_pmc_WRITE_SIZE cfg4 10 "round 4 last build: one resident batch of the headline job, 10 planes of the 32x32 grid, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py --workload cfg4 --planes 20 --steps 1 --warmup 1 --no-cpu-baseline" $O/pmc_traffic_cfg4_batch.json

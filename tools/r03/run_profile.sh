#!/bin/bash
# round 3: rocprofv3 kernel stats of the bench + PMC traffic passes on HEAD (separate runs, --kernel-trace only beside --pmc;
# the program directly after --)
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic"
S="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic"
export SQ_BENCH_NO_REFERENCE_JOB=1
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_tcc
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B > $O/bench_under_rocprof.json 2> $O/prof.err || { echo rocprof failed; tail -5 $O/prof.err; exit 1; }
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv; head -8 $O/kernel_stats.csv; cat $O/bench_under_rocprof.json
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $S > $O/pmc_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 $O/pmc_fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $S > $O/pmc_write.log 2>&1 || { echo pmc write failed; tail -5 $O/pmc_write.log; exit 1; }
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write cfg3 40 "round 3 last build (queue walk with the explicit LDS wait): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py --steps 2 --warmup 1 --no-cpu-baseline" $O/pmc_traffic_latest.json
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_sq -- $S > $O/pmc_sq.log 2>&1 || { echo pmc sq failed; tail -5 $O/pmc_sq.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- $S > $O/pmc_tcc.log 2>&1 || { echo pmc tcc failed; tail -5 $O/pmc_tcc.log; exit 1; }
python - <<'PY'
import csv, glob, json, os
out = {}
for d in ('gpurun_out/r3/pmc_sq', 'gpurun_out/r3/pmc_tcc'):
    fs = glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True)
    f = max(fs, key=os.path.getmtime)
    acc = {}
    for r in csv.DictReader(open(f)):
        if 'fuse_overwrite' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            out['vgpr'] = r.get('VGPR_Count', r.get('Arch_VGPR_Count'))
    out.update({k: sum(v) / len(v) for k, v in acc.items()})
json.dump(out, open('gpurun_out/r3/pmc_sq_tcc.json', 'w'), indent=1)
print(json.dumps(out))
PY
# the un-profiled bench line of the same box, full form (with the cpu baseline and the headline job)
unset SQ_BENCH_NO_REFERENCE_JOB
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc $?"; cat $O/bench_full.json
rm -rf $O/prof $O/pmc_fetch/*/*trace* $O/pmc_write/*/*trace*

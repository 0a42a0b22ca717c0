#!/bin/bash
# round-2 run 6: full GPU suite; rocprofv3 kernel stats + PMC passes of the bench (plane-group kernel)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests6.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r2_tests6.log
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
S="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/r2_prof gpurun_out/r2_pmc_*
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof -- $B > gpurun_out/r2_bench_under_rocprof.json 2> gpurun_out/r2_prof.err || { echo rocprof failed; tail -5 gpurun_out/r2_prof.err; exit 1; }
cat gpurun_out/r2_bench_under_rocprof.json
f=$(find gpurun_out/r2_prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r2_kernel_stats.csv; cat gpurun_out/r2_kernel_stats.csv
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_pmc_fetch -- $S > gpurun_out/r2_pmc_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 gpurun_out/r2_pmc_fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_pmc_write -- $S > gpurun_out/r2_pmc_write.log 2>&1 || { echo pmc write failed; tail -5 gpurun_out/r2_pmc_write.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/r2_pmc_sq -- $S > gpurun_out/r2_pmc_sq.log 2>&1 || { echo pmc sq failed; tail -5 gpurun_out/r2_pmc_sq.log; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/r2_pmc_tcc -- $S > gpurun_out/r2_pmc_tcc.log 2>&1 || { echo pmc tcc failed; tail -5 gpurun_out/r2_pmc_tcc.log; }
python tools/pmc_traffic.py gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write cfg3 40 "plane-group kernel (round 2), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py --steps 2 --warmup 1" && cp profiles/pmc_traffic_latest.json gpurun_out/r2_pmc_traffic_latest.json
python - <<'PY'
import csv, glob, json
out = {}
for d in ('gpurun_out/r2_pmc_sq', 'gpurun_out/r2_pmc_tcc'):
    fs = glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True)
    if not fs: continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if 'fuse_overwrite' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    out.update({k: sum(v) / len(v) for k, v in acc.items()})
json.dump(out, open('gpurun_out/r2_pmc_sq_tcc.json', 'w'), indent=1)
print(json.dumps(out))
PY

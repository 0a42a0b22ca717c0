#!/bin/bash
# SQ / TCC counters of the final config-3 launch (separate --pmc passes, kernel trace only)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
export SQ_BENCH_NO_REFERENCE_JOB=1
S="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/r2_pmc_sq gpurun_out/r2_pmc_tcc
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/r2_pmc_sq -- $S > gpurun_out/r2_pmc_sq.log 2>&1 || { echo pmc sq failed; tail -5 gpurun_out/r2_pmc_sq.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/r2_pmc_tcc -- $S > gpurun_out/r2_pmc_tcc.log 2>&1 || { echo pmc tcc failed; tail -5 gpurun_out/r2_pmc_tcc.log; exit 1; }
python - <<'PY'
import csv, glob, json, os
out = {}
for d in ('gpurun_out/r2_pmc_sq', 'gpurun_out/r2_pmc_tcc'):
    fs = glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True)
    f = max(fs, key=os.path.getmtime)
    acc = {}
    for r in csv.DictReader(open(f)):
        if 'fuse_overwrite' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    out.update({k: sum(v) / len(v) for k, v in acc.items()})
json.dump(out, open('gpurun_out/r2_pmc_sq_tcc_final.json', 'w'), indent=1)
print(json.dumps(out))
PY

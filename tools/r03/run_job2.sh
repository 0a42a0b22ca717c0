#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
for mode in "" "--host-plan"; do
  SQ_BENCH_BREAKDOWN=1 timeout -k 10 400 python bench.py --workload cfg4 --planes 25 --batch 10 --steps 6 --warmup 2 --no-cpu-baseline $mode > $O/job2$mode.json 2> $O/job2$mode.err; echo "rc $?"
  grep "\[bench\]" $O/job2$mode.err | cut -c1-400
  python - <<PY
import json
d=json.loads(open('$O/job2$mode.json').read().strip().splitlines()[-1])
print('$mode', {k:d[k] for k in ('value','ms_per_step','first_job_ms','steady_job_ms','host_ms_per_job_rank0')}, d['roofline']['frac'])
PY
done

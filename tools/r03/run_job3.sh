#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_register_gpu.py tests/test_stitcher_gpu.py tests/test_distributed_gpu.py -m gpu -x -q > $O/job3_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/job3_tests.log; [ $rc = 0 ] || exit 1
SQ_BENCH_BREAKDOWN=1 timeout -k 10 400 python bench.py --workload cfg4 --planes 25 --batch 10 --steps 6 --warmup 2 --no-cpu-baseline > $O/job3.json 2> $O/job3.err; echo "rc $?"
grep "\[bench\]" $O/job3.err | cut -c1-400
python - <<PY
import json
d=json.loads(open('$O/job3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','first_job_ms','steady_job_ms','host_ms_per_job_rank0')}, d['roofline']['frac'], d['roofline']['launch_ms'])
PY

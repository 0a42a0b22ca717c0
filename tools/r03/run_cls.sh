#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_fuse_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/cls_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $O/cls_tests.log; [ $rc = 0 ] || exit 1
SQ_BENCH_NO_REFERENCE_JOB=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $O/cls_bench.json 2> $O/cls_bench.err; echo "rc $?"
python - <<PY
import json
d=json.loads(open('$O/cls_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['parity'])
PY

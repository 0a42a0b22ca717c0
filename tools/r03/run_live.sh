#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
t0=$(date +%s)
SQ_BENCH_NO_REFERENCE_JOB=1 timeout -k 10 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/live.json 2> $O/live.err; echo "rc $? in $(( $(date +%s) - t0 )) s"
tail -3 $O/live.err | cut -c1-400
python - <<PY
import json
d=json.loads(open('$O/live.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline'])
PY

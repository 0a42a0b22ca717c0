#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
for rs in side main side main; do
  SQ_BENCH_NO_REFERENCE_JOB=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --registration-stream $rs > $O/regstream_$rs.json 2> $O/regstream_$rs.err; echo "rc $?"
  python - <<PY
import json
d=json.loads(open('$O/regstream_$rs.json').read().strip().splitlines()[-1])
print('$rs', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['parity'].get('shift_rmse_px'))
PY
done
